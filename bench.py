#!/usr/bin/env python
"""bench.py -- columns/sec (SW+LW) of the radiation() hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W`` (N>1 is launched by the driver through
``python -m torch.distributed.run``).  One "step" = one pass of radiation() over one batch of
synthetic IFS-shaped columns that is already resident in HBM.  At N=1 the workload is BASELINE.json
configs[1]: 100 000 clear-sky columns, 137 levels, ecCKD-32 SW+LW, homogeneous solver, double
precision.  Columns shard across ranks with NO data-path collective: the path has no exchange step and
every rank keeps the fluxes of the columns it owns, as the ranks of a host model do (weak scaling:
the per-GPU batch is fixed).  ``--gather`` additionally gathers the flux profiles on rank 0 every step
(RCCL), which is what an offline driver writing one output file would need.

Prints ONE JSON line on rank 0 with the contract's keys plus:
  "roofline":     dominant kernel's algorithmic bytes / its HIP-event duration vs the 8 TB/s HBM peak
  "cpu_baseline": the oracle (plain-C restatement, OpenMP over column blocks like the reference's
                  driver) timed on this box's host cores on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_column(config, nlev, which, clear_sky):
    """SURVEY.md section 8(d) figure "A", split per fused kernel: stage-interface arrays (each written
    once by its producer stage and read once by its consumer) + the compulsory inputs/outputs of the
    columns.  `which` is 'sw' or 'lw'.  W = 8 bytes."""
    W = 8
    c = 0 if clear_sky else (1 if config.do_clouds and config.i_solver_sw != 0 else 0)   # c=0 for config 2
    if which == "sw":
        a = 1 if config.use_aerosols else 0
        stage = 2 * W * nlev * (config.n_g_sw * (2 + a) + c * 3 * config.n_bands_sw)
        n_gas = (len(config.gas_optics_sw.single_gas) - 1) if config.rrtmg is None else 8   # RRTMG SW reads h2o,co2,o3,ch4,o2 (+n2o, 2 unused): count 8
        inputs = 2 * (nlev + 1) + n_gas * nlev + 1 + 2 * 6
        outputs = (6 if config.do_clear else 3) * (nlev + 1) + 7 * config.n_g_sw
    else:
        s = 1 if config.do_lw_cloud_scattering else 0
        stage = 2 * W * nlev * (config.n_g_lw * (1 + (nlev + 1) / nlev) + c * config.n_bands_lw * (1 + 2 * s))
        n_gas = (len(config.gas_optics_lw.single_gas) - 1) if config.rrtmg is None else 8   # h2o,co2,o3,n2o,co,ch4,o2 + cfc11/12/22/ccl4 folded: 8 of the arrays the LW bands read per layer on average
        inputs = 2 * (nlev + 1) + n_gas * nlev + 1 + 2
        outputs = (4 if config.do_clear else 2) * (nlev + 1) + (nlev + 1) + 4 * config.n_g_lw
    if c:
        inputs += 5 * nlev + (nlev - 1)
    if config.use_aerosols:
        inputs += 12 * nlev + nlev
    return stage + W * (inputs + outputs)


def first_columns(inputs, n):
    """The first n columns of a make_columns() result (column axis is the last one)."""
    import copy
    ncol, nlev, sl, th, gas, cloud, aer = inputs
    n = min(n, ncol)
    out = [copy.copy(o) for o in (sl, th, gas, cloud, aer)]
    for obj in out:
        if obj is None:
            continue
        for k, v in list(vars(obj).items()):
            if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == ncol:
                setattr(obj, k, np.ascontiguousarray(v[..., :n]))
    return (n, nlev, *out)


def cpu_baseline(config, workload, sample, seconds_target=12.0):
    """Time the oracle on a bounded sample (the first 2048 columns of the timed batch, repeated)."""
    from ecrad_amd.interface import Radiation
    from ecrad_amd.synthetic import BENCH_CONFIGS, make_columns
    from ecrad_amd.types import Flux
    from oracle import pyoracle
    pyoracle.build()
    nthreads = pyoracle.lib().ecrad_oracle_max_threads()
    blocked = pyoracle.make_blocked_backend(nblocksize=32, nthreads=nthreads)
    kind, what = "port", "oracle/ (plain C, "
    if config.rrtmg is not None:
        # RRTMG: the gas optics are the reference's own ifsrrtm routines (oracle/_ref, compiled from the
        # reference's sources), the rest the C restatement -- still a "port" as a whole
        if not pyoracle.have_ref_rrtm():
            raise RuntimeError("oracle/_ref/libecrad_refrrtm.so is missing (oracle/build_ref_rrtm.sh)")
        blocked = pyoracle.make_rrtmg_backend(config, inner=blocked, nthreads=int(nthreads))
        what = "the reference's ifsrrtm gas-optics routines (oracle/_ref, blocks of 4 columns on a thread pool) + oracle/ (plain C, "
    rad = Radiation(config, backend=blocked)
    ncol, nlev, sl, th, gas, cloud, aer = sample
    nsample = ncol
    flux = Flux.allocate(config, ncol, nlev)
    rad.radiation(ncol, nlev, 1, ncol, sl, th, gas, cloud, aer, flux)       # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        rad.radiation(ncol, nlev, 1, ncol, sl, th, gas, cloud, aer, flux)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or reps >= 200:
            break
    out = {"value": nsample * reps / dt, "unit": "columns/s", "cores": int(nthreads), "kind": "port",
           "sample": f"{nsample} columns x {reps} repeats of the same synthetic workload, {what}"
                     f"OpenMP over blocks of 32 columns as in driver/ecrad_driver.F90:348), {dt:.1f} s"}
    return out, flux


def main():
    os.environ.setdefault("GFORTRAN_UNBUFFERED_ALL", "1")      # (see the cpu_baseline leg below)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncol", type=int, default=100000, help="columns per GPU per step")
    ap.add_argument("--workload", default="clear_homogeneous_ecckd32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="at N>1 also gather the flux profiles on rank 0 every step (what an offline driver writing "
                         "one output file would do); off by default: the path has no exchange step, every rank "
                         "keeps the columns it owns")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes", file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from ecrad_amd.cases import make_config
    from ecrad_amd.device import DeviceCase
    from ecrad_amd.interface import Radiation
    from ecrad_amd.parallel import gather_profiles, pack_profiles
    from ecrad_amd.synthetic import BENCH_CONFIGS, make_columns
    from ecrad_amd.types import Flux

    spec = dict(BENCH_CONFIGS[args.workload])
    clear_sky = spec.pop("clear_sky")
    sw_solver = spec.pop("sw_solver")
    is_rrtmg = bool(spec.pop("rrtmg", False))
    if is_rrtmg:
        from ecrad_amd.cases import make_config_rrtmg
        config = make_config_rrtmg(sw_solver, **spec)
    else:
        config = make_config(sw_solver, **spec)
    rad = Radiation(config, backend="hip", device_id=local_rank)
    stream = torch.cuda.current_stream()
    rad.lib.ecrad_hip_set_stream(rad.handle, C.c_void_p(stream.cuda_stream))

    # weak scaling: every rank owns args.ncol columns of the global batch world*args.ncol
    ncol, nlev, sl, th, gas, cloud, aer = make_columns(config, args.ncol, clear_sky, first_column=rank * args.ncol)
    cpu_sample = first_columns((ncol, nlev, sl, th, gas, cloud, aer), 2048) if world == 1 and not args.no_cpu_baseline else None
    flux = Flux.allocate(config, ncol, nlev)
    case = DeviceCase(config, ncol, nlev, sl, th, gas, cloud, aer, flux, device=f"cuda:{local_rank}")
    profile_names = [n for n in ("lw_up", "lw_dn", "sw_up", "sw_dn", "sw_dn_direct", "lw_up_clear", "lw_dn_clear",
                                 "sw_up_clear", "sw_dn_clear", "sw_dn_direct_clear", "lw_derivatives")
                     if n in case.flux_tensors]
    do_gather = world > 1 and args.gather
    fraction0 = case.tensors["cloud_fraction"].clone() if "cloud_fraction" in case.tensors else None

    def step():
        if fraction0 is not None:      # radiation() crops cloud%fraction in place: restore the input
            case.tensors["cloud_fraction"].copy_(fraction0)
        st = rad.lib.ecrad_hip_radiation(rad.handle, ncol, nlev, 1, ncol, C.byref(case.inputs), C.byref(case.flux))
        if st != 0:
            raise RuntimeError(rad.lib.ecrad_hip_last_error(rad.handle).decode())
        if do_gather:
            packed = pack_profiles(case.flux_tensors, profile_names)
            gather_profiles(packed, [ncol] * world, dst=0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    stage_ms = {"lw": [], "sw": [], "prep": [], "post": []}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # per-stage HIP-event durations are read after the loop for the LAST step only (reading them
        # earlier would synchronise the stream inside the timed region)
    barrier()
    elapsed = time.perf_counter() - t0
    ms = C.c_double()
    for k, name in ((0, "prep"), (1, "lw"), (2, "sw"), (3, "post")):
        rad.lib.ecrad_hip_last_stage_ms(rad.handle, k, C.byref(ms))
        stage_ms[name].append(ms.value)
    # a few extra untimed steps to average the per-kernel duration
    for _ in range(3):
        step()
        torch.cuda.synchronize()
        for k, name in ((1, "lw"), (2, "sw")):
            rad.lib.ecrad_hip_last_stage_ms(rad.handle, k, C.byref(ms))
            stage_ms[name].append(ms.value)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def measured_traffic(kernel_prefix):
        """HBM bytes per launch of the dominant kernel from the newest committed PMC summary
        (profiles/*_traffic.json, written by tools/profile.sh + tools/summarize_prof.py from separate
        --pmc FETCH_SIZE / WRITE_SIZE passes of this same command), or None if none matches this
        workload and column count."""
        import glob
        best = None
        # (sorted by name: the tags are r<round>_<letter>, and file times do not survive a checkout)
        for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_traffic.json"))):
            try:
                d = json.load(open(f))
            except Exception:
                continue
            if d.get("columns") != ncol or d.get("workload") != args.workload:
                continue
            for k, v in d.get("traffic_bytes_per_launch", {}).items():
                if k.startswith(kernel_prefix):
                    best = {"bytes_per_launch": v, "source": "profiles/" + os.path.basename(f)}
        return best

    if rank == 0:
        total_cols = world * ncol * args.steps
        value = total_cols / elapsed
        dom = "sw" if np.mean(stage_ms["sw"]) >= np.mean(stage_ms["lw"]) else "lw"
        dom_ms = float(np.mean(stage_ms[dom]))
        a_bytes = algorithmic_bytes_per_column(config, nlev, dom, clear_sky)
        achieved = a_bytes * ncol / (dom_ms * 1e-3) / 1e9
        dom_kernel = f"{dom}_ica_kernel" if sw_solver != "Tripleclouds" else f"{dom}_tc_kernel"
        traffic = measured_traffic(dom_kernel)
        # spectra wider than 64 g-points run as several launches of the kernel (api.hip: chunk_lanes); the stage time
        # and the algorithmic bytes cover all of them, so the per-launch PMC figure is scaled by the launch count
        ng_dom = config.n_g_sw if dom == "sw" else config.n_g_lw
        launches = 1
        if ng_dom > 64:
            pads = {n: -(-ng_dom // n) * n for n in (64, 32, 16)}
            ok = [n for n in (64, 32, 16) if (pads[n] - ng_dom) * 100 <= 15 * ng_dom]
            best = ok[0] if ok else min((64, 32, 16), key=lambda n: (pads[n], -n))
            launches = pads[best] // best
        if traffic:
            traffic["bytes_per_launch"] *= launches
        out = {
            "metric": "columns/sec (SW+LW) at 137 lev, " + ("RRTMG 140/112" if is_rrtmg else "ecCKD-32"), "value": value, "unit": "columns/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "columns_per_gpu_per_step": ncol, "nlev": nlev,
                       "n_g_sw": config.n_g_sw, "n_g_lw": config.n_g_lw, "sw_solver": sw_solver,
                       "aerosols": bool(config.use_aerosols), "clouds": not clear_sky,
                       "parallelism": f"columns sharded over {world} GPU(s)" + (", flux profiles gathered on rank 0" if do_gather else "")},
            "roofline": {"bound": "hbm", "kernel": dom_kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic["bytes_per_launch"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "launches_per_step": launches, "algorithmic_bytes": a_bytes * ncol, "algorithmic_bytes_per_column": a_bytes, "kernel_ms": dom_ms,
                         "stage_ms": {k: float(np.mean(v)) for k, v in stage_ms.items() if v}},
        }
        if world == 1 and not args.no_cpu_baseline:
            # (the reference's RRTMG set-up routines print to Fortran unit 6: keep stdout for the JSON line)
            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            try:
                out["cpu_baseline"], oracle_flux = cpu_baseline(config, args.workload, cpu_sample)
            finally:
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
            # the oracle's sample is the first columns of this rank's batch: check the timed configuration
            # against it (outside the timed region; the oracle is the checker, never the thing measured)
            worst, nchk = 0.0, oracle_flux.ncol
            for name, t in case.flux_tensors.items():
                ref = oracle_flux.arrays.get(name)
                if ref is None:
                    continue
                got = t.cpu().numpy()
                got = got[..., :nchk] if got.shape[-1] == ncol else got[:nchk]
                scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max() + 1e-300)
                worst = max(worst, float(np.max(np.abs(got - ref) / scale)))
            out["parity"] = {"max_rel_diff_vs_oracle": worst, "columns_checked": int(nchk), "tolerance": 1e-6}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
