#!/usr/bin/env python
"""bench.py -- columns/sec (SW+LW) of the radiation() hot path on MI355X.

Contract: ``python bench.py --gpus N --steps K --warmup W``.  For N>1 it runs one process per GPU: either the driver
launches it through ``python -m torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or, started
plainly, it starts its own N ranks (ecrad_amd/parallel.py: launch_ranks) and exits with their status.  One "step" = one pass of radiation() over one batch of
synthetic IFS-shaped columns that is already resident in HBM.  The headline workload (``value``) is
BASELINE.json configs[1]: 100 000 clear-sky columns per GPU, 137 levels, ecCKD-32 SW+LW, homogeneous
solver, double precision.  Columns shard across ranks with NO data-path collective: the path has no
exchange step and every rank keeps the fluxes of the columns it owns, as the ranks of a host model do
(weak scaling: the per-GPU batch is fixed).  At N>1 the run ALSO times the same steps followed by the
one collective an offline driver writing a single file needs -- the RCCL gather of the flux profiles
on rank 0 -- and reports it as ``value_with_gather``.

Prints ONE compact JSON line (< 4 KB, strict JSON: compact_line) LAST on rank 0's standard output -- the contract's keys, `roofline`,
`cpu_baseline`, a parity summary, one short record per extra workload -- and writes the FULL record, which holds everything listed
below, to gpurun_out/bench_detail.json (emit).  The full record has the contract's keys plus:
  "roofline":     dominant stage's algorithmic bytes / its HIP-event duration vs the 8 TB/s HBM peak
  "cpu_baseline": the oracle (plain-C restatement, OpenMP over column blocks like the reference's
                  driver) timed on this box's host cores on a bounded sample of the same workload
  "parity":       the timed configuration checked against the oracle on EVERY timed column (batches of up to 125 000
                  columns; the first 16 384 of larger ones); a failure nulls ``value`` and makes the exit status non-zero
  "end_to_end_host": the same call through ECRAD_MEM_HOST pointers (copy-in, kernels and copy-out of column tiles pipelined on
                  three streams; PCIe-inclusive), next to the ceiling the measured PCIe rates set for its bytes per column
  "small_blocks": (N=1, default run) 16 host threads calling radiation() on blocks of 80 columns of host arrays at once -- the
                  reference driver's OpenMP loop over blocks, its test namelist's nblocksize -- served side by side by the
                  library's pool of contexts; the same blocks from one thread beside it
`--gpus N --threads-per-process T` is a different mode: ONE process, T host threads, blocks of --block-columns columns of host
arrays spread over N GPUs by the pool of contexts, no MPI and no torch.distributed (pool_mode).
  "workloads":    (N=1, default run) the other BASELINE configurations on this GPU, each with its own
                  value / ms_per_step / roofline / cpu_baseline / parity:
                  tripleclouds_ecckd32 (north-star shape), mcica_rrtmg (configs[2]), tripleclouds_ecckd64 at
                  1 250 000 columns (the per-GPU shard of configs[3]), spartacus_ecckd32_sp at 100 000 and at 1 250 000
                  columns (the per-GPU shard of configs[4])
"""
from __future__ import annotations

import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def progress(msg):
    """One line per phase into gpurun_out/bench_progress.log (a FILE: the driver's record is a bounded tail of stdout and stderr):
    where a run was when something outside Python ended it (a GPU memory fault aborts the process without a traceback)."""
    try:
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_progress.log"), "a") as f:
            f.write("%.3f pid %d %s\n" % (time.time(), os.getpid(), msg))
            f.flush()
            os.fsync(f.fileno())
    except OSError:
        pass

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_READ_GBS = HBM_COPY_GBS = None      # ecrad_hip_hbm_rates: pure read / copy, 16 bytes per lane, four requests in flight (2 x 1 GiB)
HBM_TRIAD_GBS = None       # measured on THIS box when the first workload is set up (ecrad_hip_hbm_triad: a = b + s c, 3 x 1 GiB)
FP64_PEAK_TFLOPS = 78.6    # MI355X vector FP64 (MI355X_MICROARCH.md quotes the FP32 vector peak, 157.3 = 2 x this)
PARITY_TOLERANCE = 1.0e-6  # BASELINE.json north_star: fluxes within 1e-6 relative of the CPU reference
EXTRA_WORKLOADS = (("tripleclouds_ecckd32", 100000), ("mcica_ecckd32", 100000), ("mcica_rrtmg", 100000), ("tripleclouds_ecckd64", 1250000),
                   ("spartacus_ecckd32_sp", 100000), ("spartacus_ecckd32_sp", 1250000))
CHUNK_COLUMNS = 125000     # synthetic columns are generated and uploaded this many at a time (bounds host memory)


def algorithmic_bytes_per_column(config, nlev, which, clear_sky):
    """SURVEY.md section 8(d) figure "A", split per fused stage: stage-interface arrays (each written
    once by its producer stage and read once by its consumer) + the compulsory inputs/outputs of the
    columns.  `which` is 'sw' or 'lw'.  W = 8 bytes."""
    from ecrad_amd.config import ISolverSpartacus
    # (the SPARTACUS solvers in single precision pass 4-byte words between their stages: SURVEY 8(d), config 5)
    W = 4 if (getattr(config, "i_precision", 0) == 1 and ISolverSpartacus in (config.i_solver_sw, config.i_solver_lw)) else 8
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


def algorithmic_flops_per_column(config, nlev, clear_sky):
    """SURVEY.md section 8(d), "ALGORITHMIC flops per column (estimate, count exp ~ 40, sqrt/div ~ 15 flop-equivalents)":
    gas optics ~ 80 per (g, level) and spectrum, shortwave two-stream + adding ~ 250, longwave no-scatter ~ 100
    => 2.2 MFLOP for the clear-sky ecCKD-32 column; Tripleclouds ~ 3x that, McICA ~ 2x (+ the serial generator).
    An estimate by the survey's own count, used for `roofline.fp64_fraction` only."""
    from ecrad_amd.config import ISolverMcICA, ISolverTripleclouds, ISolverSpartacus
    base = nlev * (config.n_g_sw * (80 + 250) + config.n_g_lw * (80 + 100))
    if clear_sky or not config.do_clouds:
        return float(base)
    return float(base * {ISolverTripleclouds: 3.0, ISolverSpartacus: 3.0, ISolverMcICA: 2.0}.get(config.i_solver_sw, 1.0))


def build_config(workload):
    """(config, clear_sky, description dict) of a named workload of ecrad_amd/synthetic.py: BENCH_CONFIGS."""
    from ecrad_amd.cases import make_config, make_config_rrtmg
    from ecrad_amd.synthetic import BENCH_CONFIGS
    spec = dict(BENCH_CONFIGS[workload])
    clear_sky = spec.pop("clear_sky")
    sw_solver = spec.pop("sw_solver")
    is_rrtmg = bool(spec.pop("rrtmg", False))
    config = make_config_rrtmg(sw_solver, **spec) if is_rrtmg else make_config(sw_solver, **spec)
    return config, clear_sky, {"sw_solver": sw_solver, "rrtmg": is_rrtmg}


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


def oracle_backend(config, nthreads=None, fma=False):
    """The checker / CPU baseline: oracle/ with OpenMP over blocks of 32 columns (driver/ecrad_driver.F90:348); for
    RRTMG its gas optics are the reference's own ifsrrtm routines (oracle/_ref).  Returns (backend, description, threads).
    fma: the same restatement compiled with floating-point contraction -- never the parity reference, only the measure of
    how far the reference's own formulas move when nothing but the rounding of a*b+c changes."""
    from oracle import pyoracle
    pyoracle.build()
    nthreads = int(nthreads or pyoracle.lib().ecrad_oracle_max_threads())
    single = getattr(config, "i_precision", 0) == 1     # the single-precision build of the SPARTACUS restatement (PARKIND1_SINGLE)
    if fma:
        backend = (pyoracle.make_variant_backend("sp_fma", nblocksize=32, nthreads=nthreads) if single
                   else pyoracle.make_fma_variant_backend(nblocksize=32, nthreads=nthreads))
    elif single:
        backend = pyoracle.make_variant_backend("sp", nblocksize=32, nthreads=nthreads)
    else:
        backend = pyoracle.make_blocked_backend(nblocksize=32, nthreads=nthreads)
    what = "oracle/ (plain C, -O3, " + ("SPARTACUS solvers in single precision, " if single else "")
    from ecrad_amd.config import IGasModelIFSRRTMG
    if IGasModelIFSRRTMG in (config.i_gas_model_sw, config.i_gas_model_lw):
        if not pyoracle.have_ref_rrtm():
            raise RuntimeError("oracle/_ref/libecrad_refrrtm.so is missing (oracle/build_ref_rrtm.sh)")
        backend = pyoracle.make_rrtmg_backend(config, inner=backend, nthreads=nthreads)
        what = ("gas optics: the reference's ifsrrtm routines (oracle/_ref, \"reference\"; blocks of 8 columns on a thread pool) + "
                "everything else: oracle/ (plain C, -O3, ")
    return backend, what, nthreads


def oracle_flux_of(config, inputs, fma=False):
    """Fluxes of the oracle for make_columns()-style inputs (all their columns)."""
    from ecrad_amd.interface import Radiation
    from ecrad_amd.types import Flux
    backend, _, _ = oracle_backend(config, fma=fma)
    rad = Radiation(config, backend=backend)
    ncol, nlev, sl, th, gas, cloud, aer = inputs
    flux = Flux.allocate(config, ncol, nlev)
    rad.radiation(ncol, nlev, 1, ncol, sl, th, gas, cloud, aer, flux)
    return flux


def columns_of(inputs, c0, n):
    """Columns c0 .. c0+n-1 of a make_columns() result (column axis is the last one)."""
    import copy
    ncol, nlev, sl, th, gas, cloud, aer = inputs
    n = min(n, ncol - c0)
    out = [copy.copy(o) for o in (sl, th, gas, cloud, aer)]
    for obj in out:
        if obj is None:
            continue
        for k, v in list(vars(obj).items()):
            if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == ncol:
                setattr(obj, k, np.ascontiguousarray(v[..., c0:c0 + n]))
    return (n, nlev, *out)


def cpu_baseline(config, sample, seconds_target=10.0, workload_name=None):
    """Time the oracle on a bounded sample (the first columns of the timed batch, repeated for ~10 s)."""
    from ecrad_amd.interface import Radiation
    from ecrad_amd.types import Flux
    backend, what, nthreads = oracle_backend(config)
    rad = Radiation(config, backend=backend)
    ncol, nlev, sl, th, gas, cloud, aer = sample
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
    out = {"value": ncol * reps / dt, "unit": "columns/s", "cores": nthreads, "kind": "port",
           "sample": f"{ncol} columns ({ncol // 32} blocks of 32 for {nthreads} threads) x {reps} repeats of the same synthetic "
                     f"workload, {what}OpenMP over blocks of 32 columns as in driver/ecrad_driver.F90:348), {dt:.1f} s"}
    try:
        comp = reference_leaf_component(config, rad, sample, nthreads)
        if comp:
            out["components"] = comp
    except Exception as e:       # the reference-code component is additional information: never take the line down
        out["components"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        ref = reference_executable_component(workload_name, config, sample, nthreads) if workload_name else None
        if ref:
            # north_star: "next to the reference Fortran/OpenMP path timed on the same box's host cores".  Where the reference's
            # own executable exists it IS the baseline (`kind: "reference"`); the oracle's figure stays beside it as a component.
            port = {k: out[k] for k in ("value", "unit", "cores", "kind", "sample")}
            comp = out.get("components", {})
            out = dict(ref, components=dict(comp, port=port))
    except Exception as e:
        out.setdefault("components", {})["reference_executable"] = {"error": f"{type(e).__name__}: {e}"}
    return out, flux


def reference_leaf_component(config, rad, sample, nthreads, seconds_target=4.0):
    """The part of the CPU baseline that can run as the REFERENCE's own Fortran on this box: the solver stage of the
    clear-sky homogeneous workload (two-stream coefficients + adding method, SW and LW) by the reference's leaf routines,
    compiled unmodified into oracle/_ref/libecrad_refleaf.so, in the reference's calling order with OpenMP over blocks of
    32 columns (oracle/ref_leaf_wrappers.F90: ref_clear_sky_solvers; pinned by tests/test_oracle_vs_ref_leaf.py).  The
    stage arrays it works on come from the oracle's optics stage.  Only that workload has a solver stage made of leaf
    routines alone; the gas-optics stage of ecCKD needs the reference's netCDF-dependent modules and stays "port"."""
    from oracle import pyoracle
    from ecrad_amd.config import ISolverHomogeneous
    if not pyoracle.have_ref_leaf() or config.use_aerosols or config.i_solver_sw != ISolverHomogeneous or config.rrtmg is not None:
        return None
    from ecrad_amd.interface import build_inputs_struct
    ncol, nlev, sl, th, gas, cloud, aer = sample
    if cloud is not None and float(np.max(cloud.fraction)) > 0.0:
        return None
    n0 = min(ncol, 1024)                               # the oracle's optics stage is serial: a small sample, then tiled
    sub = first_columns(sample, n0)
    cin, keep = build_inputs_struct(config, *sub)
    stage = pyoracle.optics(config, rad.cconfig, n0, nlev, 1, n0, cin)
    reps_t = max(1, 16384 // n0)
    stage = {k: np.ascontiguousarray(np.concatenate([v] * reps_t, axis=0)) for k, v in stage.items()
             if k in ("od_sw", "ssa_sw", "g_sw", "incoming_sw", "sw_albedo_diffuse", "sw_albedo_direct", "od_lw", "planck_hl", "lw_emission", "lw_albedo")}
    mu0 = np.concatenate([sub[2].cos_sza] * reps_t)
    n = n0 * reps_t
    os.environ.setdefault("OMP_NUM_THREADS", str(nthreads))
    pyoracle.ref_clear_sky_solvers(stage, mu0, 32)      # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        pyoracle.ref_clear_sky_solvers(stage, mu0, 32)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or reps >= 100:
            break
    return {"solver_stage": {"value": n * reps / dt, "unit": "columns/s", "cores": nthreads, "kind": "reference",
                             "what": "two-stream coefficients + adding method, SW and LW, of the clear-sky homogeneous solvers: the reference's own "
                                     "calc_two_stream_gammas_sw, calc_reflectance_transmittance_sw, adding_ica_sw, calc_no_scattering_transmittance_lw, "
                                     "calc_fluxes_no_scattering_lw (amdflang -O3 -fopenmp, oracle/_ref), OpenMP over blocks of 32 columns",
                             "sample": f"{n} columns ({n0} distinct) x {reps} repeats, {dt:.1f} s"},
            "everything_else": {"kind": "port", "what": "gas optics, Planck function, albedo mapping, flux post-processing: oracle/ (the reference's "
                                                         "modules for these need netCDF and cannot be built here)"}}


REFERENCE_EXE = os.path.join(ROOT, "tests", "_build", "reference", "ecrad_ref")

# bench workload -> change_namelist.sh-style edits of tests/golden/configCY49R1_ecckd.nam (= test/ifs/configCY49R1_ecckd.nam)
REFERENCE_NAMELIST_EDITS = {
    "clear_homogeneous_ecckd32": {"sw_solver_name": '"Homogeneous"', "lw_solver_name": '"Homogeneous"', "use_aerosols": "false",
                                  "do_save_spectral_flux": "false"},
    "tripleclouds_ecckd32": {"do_save_spectral_flux": "false"},
}


def reference_executable_component(name, config, sample, nthreads, seconds_target=8.0):
    """The reference's OWN offline executable timed on this box's host cores (north_star: "next to the reference Fortran /
    OpenMP path timed on the same box's host cores"): tests/_build/reference/ecrad_ref is ecmwf-ifs/ecrad 1.7.1 compiled
    unmodified by tools/build_dropin.py --reference (amdflang -O3 -fopenmp) -- every line of radiation() is the reference's --
    on top of this repo's netCDF library (the image has no libnetcdff), which only serves the file reads before and the write
    after the timed loop.  Because of that library it is reported as a component next to the "port" figure, not as
    oracle/_ref.  The sample's columns are written to a netCDF file (ecrad_amd.driver.save_inputs), the driver runs them in
    blocks of 32 columns over all host threads (its own !$OMP PARALLEL DO, driver/ecrad_driver.F90:348) `nrepeat` times and
    prints its own timer (:387)."""
    import re
    import subprocess
    import tempfile
    if not os.path.exists(REFERENCE_EXE) or name not in REFERENCE_NAMELIST_EDITS:
        return None
    from ecrad_amd.driver import save_inputs
    ncol, nlev, sl, th, gas, cloud, aer = sample
    ncol_ref = min(ncol, 8192)
    sub = first_columns(sample, ncol_ref)
    with tempfile.TemporaryDirectory() as tmp:
        inp = os.path.join(tmp, "inputs.nc")
        save_inputs(inp, config, *sub[2:])
        text = open(os.path.join(ROOT, "tests", "golden", "configCY49R1_ecckd.nam")).read()
        head, rad = text.split("&radiation\n", 1)
        edits = dict(REFERENCE_NAMELIST_EDITS[name], directory_name=f'"{os.path.join(ROOT, "data")}"', iverbosesetup="0", iverbose="0")
        for k, v in edits.items():
            pat = re.compile(r"^(\s*)" + re.escape(k) + r"\s*=[^,\n]*,?", re.M)
            rad = pat.sub(lambda m: f"{m.group(1)}{k} = {v},", rad, count=1) if pat.search(rad) else f"{k} = {v},\n" + rad

        def run(nrepeat):
            h = re.sub(r"nrepeat\s*=\s*\d+", f"nrepeat = {nrepeat}", head)
            h = re.sub(r"nblocksize\s*=\s*\d+", "nblocksize = 32", h)
            nam = os.path.join(tmp, "config.nam")
            open(nam, "w").write(h + "&radiation\n" + rad)
            env = dict(os.environ, OMP_NUM_THREADS=str(nthreads), OMP_STACKSIZE="1G")
            p = subprocess.run(f"ulimit -s unlimited; exec {REFERENCE_EXE} {nam} {inp} {os.path.join(tmp, 'out.nc')}", shell=True,
                               capture_output=True, text=True, env=env, cwd=tmp, timeout=600)
            m = re.search(r"Time elapsed in radiative transfer:\s*([0-9.Ee+-]+)\s*seconds", p.stdout + p.stderr)
            if p.returncode != 0 or not m:
                raise RuntimeError((p.stdout + p.stderr)[-500:])
            return float(m.group(1))
        t1 = run(1)
        nrepeat = int(max(1, min(200, seconds_target / max(t1, 1e-3))))
        t = run(nrepeat)
    return {"value": ncol_ref * nrepeat / t, "unit": "columns/s", "cores": nthreads, "kind": "reference",
            "what": "the reference's own offline executable (ecmwf-ifs/ecrad 1.7.1 compiled unmodified, amdflang -O3 -fopenmp; its netCDF "
                    "reads and writes, outside the timed loop, served by this repo's netcdf module), OpenMP over blocks of 32 columns, "
                    "the driver's own timer",
            "sample": f"{ncol_ref} columns x nrepeat = {nrepeat}, {t:.1f} s"}


def measured_traffic(workload, ncol, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/*_traffic.json,
    written by tools/profile.sh + tools/summarize_prof.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes of this
    same command), or None if none matches this workload and column count."""
    best = None
    # (sorted by name: the tags are r<round>_<letter>, and file times do not survive a checkout)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("columns") != ncol or d.get("workload") != workload:
            continue
        # A spectrum wider than 64 g-points runs as launches of DIFFERENT instantiations of the kernel (chunks of 64, 32
        # and 16 lanes): the per-launch figure is the average over the launches the profile saw, weighted by their counts
        tb, nl = d.get("traffic_bytes_per_launch", {}), d.get("launches_profiled", {})
        names = [k for k in tb if k.startswith(tuple(kernel_prefix) if isinstance(kernel_prefix, (tuple, list)) else kernel_prefix)]
        if names:
            wsum = sum(nl.get(k, 1) for k in names)
            best = {"bytes_per_launch": sum(tb[k] * nl.get(k, 1) for k in names) / wsum, "source": "profiles/" + os.path.basename(f),
                    "columns_per_launch": d.get("columns_per_launch", ncol)}
    return best


def measured_valu(workload, kernel_prefix):
    """VALU-issue figures of the dominant kernel from the newest committed SQ-counter summary (profiles/*_sq.json, written by
    tools/summarize_sq.py from the `rocprofv3 --pmc` passes of tools/pmc_sq.sh over this same command), or None."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq.json"))):
        try:
            ks = json.load(open(f))["workloads"].get(workload, {})
        except Exception:
            continue
        names = [k for k in ks if k.startswith(tuple(kernel_prefix)) and ks[k].get("valu_busy") is not None]
        if names:
            k = max(names, key=lambda n: ks[n].get("valu_insts") or 0.0)
            best = {"busy": ks[k]["valu_busy"], "waiting": ks[k]["waiting"], "insts_per_wave_layer": ks[k].get("valu_per_wave_layer"),
                    "kernel": k, "source": "profiles/" + os.path.basename(f)}
    return best


class Workload:
    """One named workload resident in HBM on this rank: configuration, library handle, device arrays."""

    def __init__(self, name, ncol, rank, local_rank, want_sample):
        import torch
        from ecrad_amd.device import DeviceCase
        from ecrad_amd.interface import Radiation
        from ecrad_amd.synthetic import make_columns
        from ecrad_amd.types import Flux
        self.name, self.ncol = name, ncol
        self.config, self.clear_sky, self.desc = build_config(name)
        self.rad = Radiation(self.config, backend="hip", device_id=local_rank)
        self.rad.lib.ecrad_hip_set_stream(self.rad.handle, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        global HBM_TRIAD_GBS
        if HBM_TRIAD_GBS is None:       # once per process, before the batch occupies the memory
            gbs = C.c_double()
            if self.rad.lib.ecrad_hip_hbm_triad(self.rad.handle, C.c_size_t(1 << 30), 5, C.byref(gbs)) != 0:
                raise RuntimeError(self.rad.lib.ecrad_hip_last_error(self.rad.handle).decode())
            HBM_TRIAD_GBS = gbs.value
            global HBM_READ_GBS, HBM_COPY_GBS
            rd, cp = C.c_double(), C.c_double()
            if self.rad.lib.ecrad_hip_hbm_rates(self.rad.handle, C.c_size_t(1 << 30), 5, C.byref(rd), C.byref(cp)) != 0:
                raise RuntimeError(self.rad.lib.ecrad_hip_last_error(self.rad.handle).decode())
            HBM_READ_GBS, HBM_COPY_GBS = rd.value, cp.value
        device = f"cuda:{local_rank}"
        # weak scaling: every rank owns `ncol` columns of the global batch; generated and uploaded in chunks
        self.sample, self.host_inputs = None, None
        cases = []
        for c0 in range(0, ncol, CHUNK_COLUMNS):
            n = min(CHUNK_COLUMNS, ncol - c0)
            inputs = make_columns(self.config, n, self.clear_sky, first_column=rank * ncol + c0)
            self.nlev = inputs[1]
            if c0 == 0:
                if want_sample:
                    self.sample = first_columns(inputs, want_sample)
                if ncol <= CHUNK_COLUMNS:
                    self.host_inputs = inputs
            cases.append(DeviceCase(self.config, n, self.nlev, *inputs[2:], Flux.allocate(self.config, n, self.nlev), device=device))
        if len(cases) == 1:
            self.case = cases[0]
        else:
            self.case = DeviceCase.concatenate(self.config, cases)
            del cases
            torch.cuda.empty_cache()
        self.fraction0 = self.case.tensors["cloud_fraction"].clone() if "cloud_fraction" in self.case.tensors else None
        self.profile_names = [n for n in ("lw_up", "lw_dn", "sw_up", "sw_dn", "sw_dn_direct", "lw_up_clear", "lw_dn_clear",
                                          "sw_up_clear", "sw_dn_clear", "sw_dn_direct_clear", "lw_derivatives")
                              if n in self.case.flux_tensors]

    def step(self, gather_world=0):
        from ecrad_amd.parallel import gather_profiles, pack_profiles
        rad, case = self.rad, self.case
        if self.fraction0 is not None:      # radiation() crops cloud%fraction in place: restore the input
            case.tensors["cloud_fraction"].copy_(self.fraction0)
        st = rad.lib.ecrad_hip_radiation(rad.handle, self.ncol, self.nlev, 1, self.ncol, C.byref(case.inputs), C.byref(case.flux))
        if st != 0:
            raise RuntimeError(rad.lib.ecrad_hip_last_error(rad.handle).decode())
        if gather_world:
            packed = pack_profiles(case.flux_tensors, self.profile_names)
            if os.environ.get("ECRAD_BENCH_TEST_SHARED_GPU") == "1":
                packed = packed.cpu()       # (gloo has no device gather)
            self.gathered, _ = gather_profiles(packed, [self.ncol] * gather_world, dst=0)

    def stage_ms(self):
        ms, out = C.c_double(), {}
        for k, name in ((0, "prep"), (1, "lw"), (2, "sw"), (3, "post")):
            self.rad.lib.ecrad_hip_last_stage_ms(self.rad.handle, k, C.byref(ms))
            out[name] = ms.value
        return out

    def call_info(self):
        from ecrad_amd import abi
        info = abi.CallInfo()
        self.rad.lib.ecrad_hip_last_call_info(self.rad.handle, C.byref(info))
        return info

    def close(self):
        import torch
        self.rad.close()
        self.case = None
        self.fraction0 = None
        torch.cuda.empty_cache()


def timed_steps(w, steps, warmup, barrier, gather_world=0):
    import gc
    for _ in range(warmup):
        w.step(gather_world)
    # (the host side of a step is a few dozen microseconds of ctypes; a collection of the interpreter's garbage -- the default run
    #  has built and dropped several workloads by the time the later ones are timed -- is tens of milliseconds: not inside the region)
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            w.step(gather_world)
        barrier()
        return time.perf_counter() - t0
    finally:
        if was_enabled:
            gc.enable()


LIBRARY_GATHER_HUNG = False      # set when the library's RCCL gather did not come back: the process then leaves through os._exit after its line
LIBRARY_GATHER_FAILED = False    # set when this rank's leg ended with an error while others may be waiting in its collective: the same exit


def library_gather_leg(w, world, rank, steps, barrier, allreduce_max, timeout=180.0):
    """The same K steps with the profiles put together on rank 0 by the LIBRARY's own gather (include/ecrad_hip.h: ecrad_hip_comm_init,
    ecrad_hip_gather_profiles: RCCL directly, what a Fortran host with one rank per GPU calls) instead of torch.distributed.  Rank 0's id
    travels to the other ranks over the process group that is there anyway (an MPI host would MPI_Bcast it).  The result is compared with
    what torch's gather delivered.  Runs in a thread with a deadline: multi-rank RCCL cannot be tried on the one-GPU boxes this was built
    on (RCCL refuses two ranks on one device), so a communicator that never forms must not cost the run its line -- after `timeout`
    seconds the record says so and the process ends without waiting for the stuck thread."""
    import threading
    import torch
    box = {}

    def body():
        try:
            from ecrad_amd.parallel import library_comm_init, library_gather_profiles

            def bcast(b):
                if world == 1:
                    return b
                import torch.distributed as dist
                obj = [b]
                dist.broadcast_object_list(obj, src=0)
                return obj[0]
            names, nrows = w.profile_names, w.nlev + 1
            local = [w.case.flux_tensors[n] for n in names]
            torch.cuda.set_device(local[0].device)      # (a new thread starts on device 0: the barrier's collective must run on this rank's GPU)
            library_comm_init(w.rad, rank, world, bcast)
            assert all(t.is_contiguous() and tuple(t.shape) == (nrows, w.ncol) for t in local)
            glob = torch.zeros((len(names), nrows, world * w.ncol), dtype=torch.float64, device=local[0].device) if rank == 0 else None
            lp = [t.data_ptr() for t in local]
            gp = [glob[i].data_ptr() for i in range(len(names))] if rank == 0 else None

            def step():
                w.step(0)
                library_gather_profiles(w.rad, None, [w.ncol] * world, root=0, rank=rank, device_pointers=(lp, gp, nrows))
            step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            barrier()
            e = allreduce_max(time.perf_counter() - t0)
            rec = {"value": world * w.ncol * steps / e, "ms_per_step": 1e3 * e / steps, "how": "ecrad_hip_gather_profiles (librccl, no torch on the data path)"}
            if rank == 0:
                rec["own_share_intact"] = bool(all(torch.equal(glob[i][:, :w.ncol], local[i]) for i in range(len(names))))
                if getattr(w, "gathered", None) is not None and len(w.gathered) == world:
                    rec["same_as_torch_gather"] = bool(all(torch.equal(glob[:, :, r * w.ncol:(r + 1) * w.ncol], w.gathered[r][..., :w.ncol].to(glob.device))
                                                           for r in range(world)))
            w.rad.lib.ecrad_hip_comm_destroy(w.rad.handle)
            box["rec"] = rec
        except Exception as e:
            box["rec"] = {"error": f"{type(e).__name__}: {e}"}

    t = threading.Thread(target=body, daemon=True)
    t.start()
    t.join(timeout)
    global LIBRARY_GATHER_HUNG, LIBRARY_GATHER_FAILED
    if t.is_alive():
        LIBRARY_GATHER_HUNG = True
        return {"error": f"no completion within {timeout:.0f} s (rank {rank}); the line above it is unaffected"}
    if "error" in box["rec"] and world > 1:
        LIBRARY_GATHER_FAILED = True      # (the other ranks are stuck in the collective this rank left: no barrier with them at the end)
    return box["rec"]


def roofline_of(w, stage_ms, elapsed_per_step_s):
    """HIP events on the launch stream bracket the LW and SW stages of a call (ecrad_hip_last_stage_ms); the dominant
    one is priced against SURVEY 8(d)'s algorithmic bytes of that stage."""
    config, nlev, ncol = w.config, w.nlev, w.ncol
    dom = "sw" if stage_ms["sw"] >= stage_ms["lw"] else "lw"
    dom_ms = stage_ms[dom]
    a_dom = algorithmic_bytes_per_column(config, nlev, dom, w.clear_sky)
    a_all = sum(algorithmic_bytes_per_column(config, nlev, k, w.clear_sky) for k in ("sw", "lw"))
    achieved = a_dom * ncol / (dom_ms * 1e-3) / 1e9
    info = w.call_info()
    launches = info.launches_sw if dom == "sw" else info.launches_lw
    kernel = {"Tripleclouds": f"{dom}_tc_kernel", "SPARTACUS": f"spartacus_{dom}_kernel"}.get(w.desc["sw_solver"], f"{dom}_ica_kernel")
    # (instantiations over float tables serve the ecCKD models, the table-free StageD ones -- `double` before the second half of
    #  round 4 -- the RRTMG stage arrays: a profile of the default run holds both, e.g. sw_ica_kernel<FixedF, 32, 1, ...> and
    #  sw_ica_kernel<StageD, 64, 2, ...>)
    is_rrtmg = bool(w.desc.get("rrtmg"))
    # (the ecCKD kernels run as their FixedF instantiations -- compile-time quad counts -- for every model shipped so far)
    prefixes = (kernel + "<StageD,", kernel + "<double,") if is_rrtmg else (kernel + "<float,", kernel + "<FixedF,")
    traffic = measured_traffic(w.name, ncol, prefixes)
    valu = measured_valu(w.name, prefixes)
    whole = a_all * ncol / elapsed_per_step_s / 1e9
    extra = {}
    if w.desc["sw_solver"] == "SPARTACUS":
        # Compute-bound ("treat as FP32 vector-FMA bound"): one 9x9 (SW) / 6x6 (LW) matrix exponential per
        # g-point and cloudy layer.  Flops are ESTIMATED from the number of cloudy layers of the batch: Pade-7 with 3
        # squarings, the LU solve and the 3x3 adding algebra = 4000 (SW) / 2300 (LW) fused multiply-adds.
        frac = w.case.tensors["cloud_fraction"]
        cloudy = int((frac >= config.cloud_fraction_threshold).sum().item())
        fma = {"sw": 4000, "lw": 2300}[dom] * (config.n_g_sw if dom == "sw" else config.n_g_lw)
        flops = 2.0 * fma * cloudy
        single = getattr(config, "i_precision", 0) == 1
        peak = 157.3 if single else 78.6           # MI355X_MICROARCH.md: vector FP32 / FP64 TFLOP/s
        extra["compute"] = {"bound": "valu_fp32" if single else "valu_fp64", "estimated_flops": flops, "cloudy_layers_per_column": cloudy / ncol,
                            "achieved": flops / (dom_ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                            "frac": flops / (dom_ms * 1e-3) / 1e12 / peak}
    single = getattr(config, "i_precision", 0) == 1
    flops = algorithmic_flops_per_column(config, nlev, w.clear_sky)
    vpeak = 2 * FP64_PEAK_TFLOPS if single else FP64_PEAK_TFLOPS
    # (ECRAD_HIP_EXACT_SCRATCH=1 in the environment: the handle runs the unpacked instantiations, include/ecrad_hip.h)
    packed = w.desc["sw_solver"] in ("Cloudless", "Homogeneous", "McICA", "Tripleclouds") and os.environ.get("ECRAD_HIP_EXACT_SCRATCH", "")[:1] != "1"
    # what `kernel_ms` (HIP events around the LW / SW stage of a call) covers besides the dominant kernel named in `kernel`
    if w.desc["sw_solver"] == "SPARTACUS":
        scope = f"stage: optics_dump_kernel + spartacus_layers_kernel + spartacus_{dom}_kernel (`traffic` is the sweep kernel's own)"
    elif w.desc["sw_solver"] == "McICA":
        scope = f"stage: mcica_generator_kernel + {launches} launch(es) of {kernel}"
    else:
        scope = f"{launches} launch(es) of {kernel}"
    # Which roof binds.  `bound`/`achieved`/`peak`/`frac` price the kernel against HBM with SURVEY 8(d)'s ALGORITHMIC bytes (the
    # contract's definition); next to it the fraction of the HBM peak the kernel's COUNTED traffic makes, and the fraction of
    # cycles its SIMDs spend issuing vector instructions (SQ counters of a committed profile).  The largest of the three is the
    # roof the kernel is closest to: `binding_roof`.
    t_bytes = traffic["bytes_per_launch"] / traffic["columns_per_launch"] * ncol * launches if traffic else None
    fractions = {"hbm_algorithmic": achieved / HBM_PEAK_GBS,
                 "hbm_counter": (t_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_bytes else None,
                 "valu_issue": valu["busy"] if valu else None}
    binding = max((k for k in fractions if fractions[k] is not None), key=lambda k: fractions[k])
    return {**extra, "bound": "hbm", "binding_roof": binding, "fractions": fractions, "valu": valu, "kernel": kernel, "kernel_ms_scope": scope, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "measured_triad": HBM_TRIAD_GBS, "frac_of_measured_triad": achieved / HBM_TRIAD_GBS,
            "measured_read": HBM_READ_GBS, "measured_copy": HBM_COPY_GBS, "frac_of_measured_copy": achieved / HBM_COPY_GBS if HBM_COPY_GBS else None,
            # SURVEY 8(d)'s third number: columns/s x estimated flops per column / the vector peak of the working precision
            "fp64_fraction": ncol / elapsed_per_step_s * flops / 1e12 / vpeak,
            "flops_per_column_estimate": flops, "vector_peak_tflops": vpeak,
            # the shortwave sweep records of these solvers travel as five doubles in 32 bytes: 39 mantissa bits, rounded to
            # nearest (kernels_common.h: pack5); everything else, and all arithmetic, is binary64
            "scratch_mantissa_bits": 39 if packed else (24 if single else 53),
            # spectra wider than 64 g-points run as several launches of the kernel; the stage time and the algorithmic
            # bytes cover all of them (and, for McICA, the cloud generator that feeds them), the PMC figure is per launch
            # (scaled by columns per launch: the profile may have run the call as a different number of column tiles)
            "traffic": t_bytes,
            "traffic_source": traffic["source"] if traffic else None,
            "launches_per_step": launches * info.n_tiles, "column_tiles": info.n_tiles,
            "algorithmic_bytes": a_dom * ncol, "algorithmic_bytes_per_column": a_dom, "kernel_ms": dom_ms,
            "whole_step": {"algorithmic_bytes_per_column": a_all, "achieved": whole, "frac": whole / HBM_PEAK_GBS},
            "stage_ms": stage_ms, "work_bytes": int(info.work_bytes)}


def parity_tolerance(config):
    """1e-6 (north_star, double precision).  The single-precision SPARTACUS workload is checked against the oracle's
    single-precision build (the reference's PARKIND1_SINGLE semantics): float rounding through 9x9 matrix exponentials and
    unpivoted LU solves leaves the shortwave and the clear-sky longwave good to a few 1e-4 (the oracle's own last-bit
    sensitivity there, tests/test_hip_spartacus.py), bar 2e-3; the ALL-SKY LONGWAVE with 3-D effects is chaotic in single
    precision in the reference's own formulation (radiation_config.F90:1144 warns) and is reported as statistics only."""
    return 2.0e-3 if getattr(config, "i_precision", 0) == 1 else PARITY_TOLERANCE


def check_parity(w, oracle_flux, inputs=None):
    """The timed configuration (double precision) against the oracle on EVERY column the oracle was run on (outside the timed
    region; the oracle is the checker, never the thing measured).  No column is ever set aside: a non-finite value anywhere
    fails the check.  Names the field, column and level of the largest difference and, when that difference exceeds 1e-8,
    how far the oracle ITSELF moves at that very element when it is compiled with floating-point contraction
    (`oracle_fma_vs_plain_there`): a difference of the size of the formulas' own last-bit sensitivity is conditioning (the
    Meador-Weaver direct-beam bracket divided by 1 - (k mu0)^2), not a defect -- it is NAMED, the gate stays the tolerance."""
    worst = {"max_rel_diff_vs_oracle": 0.0, "field": None}
    worst_broadband = 0.0
    nchk = oracle_flux.ncol
    worst_col = None
    tol = parity_tolerance(w.config)
    for name, t in w.case.flux_tensors.items():
        ref = oracle_flux.arrays.get(name)
        if ref is None:
            continue
        col_last = t.shape[-1] == w.ncol
        got = (t[..., :nchk] if col_last else t[:nchk]).cpu().numpy()
        if not np.all(np.isfinite(got)):
            return {"max_rel_diff_vs_oracle": float("nan"), "field": name, "columns_checked": int(nchk), "tolerance": tol, "ok": False}
        scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max() + 1e-300)
        err = np.abs(got - ref) / scale
        idx = np.unravel_index(int(np.argmax(err)), err.shape)
        if err[idx] > worst["max_rel_diff_vs_oracle"]:
            worst = {"max_rel_diff_vs_oracle": float(err[idx]), "field": name, "index": [int(i) for i in idx]}
            worst_col = int(idx[-1] if col_last else idx[0])
            worst_local = (name, idx, col_last, float(scale[idx]))
        if not name.endswith(("_g", "_band", "_canopy")):
            worst_broadband = max(worst_broadband, float(err[idx]))
    # `max_rel_diff_broadband`: the flux profiles, derivatives and cloud cover (north_star: "fluxes within 1e-6 relative");
    # `max_rel_diff_vs_oracle`: those and every per-g-point / per-band / canopy diagnostic
    worst.update({"max_rel_diff_broadband": worst_broadband, "columns_checked": int(nchk), "columns_timed": int(w.ncol), "tolerance": tol,
                  "ok": bool(worst["max_rel_diff_vs_oracle"] <= tol)})
    if inputs is not None and worst_col is not None and worst["max_rel_diff_vs_oracle"] > 1.0e-8:
        # the block of 32 columns (the oracle's own blocking) that holds the worst element, plain and contracted
        name, idx, col_last, scale_at = worst_local
        c0 = (worst_col // 32) * 32
        blk = columns_of(inputs, c0, 32)
        a = oracle_flux_of(w.config, blk).arrays[name]
        b = oracle_flux_of(w.config, blk, fma=True).arrays[name]
        j = list(idx)
        j[-1 if col_last else 0] = worst_col - c0
        sens = float(abs(a[tuple(j)] - b[tuple(j)]) / scale_at)
        worst["oracle_fma_vs_plain_there"] = sens
        # A per-g-point diagnostic at which the reference's OWN formula moves by more than the tolerance when only the
        # rounding of a*b+c changes (the Meador-Weaver direct-beam bracket divided by 1 - (k mu0)^2, a handful of elements
        # in 10^9) cannot be held to the tolerance by anybody: named as such (`explained_by_conditioning`) if the broadband
        # fluxes hold the tolerance and the element is within 10x the tolerance and within 3x that sensitivity.  `ok` stays false.
        if (not worst["ok"] and worst_broadband <= tol and worst["field"].endswith(("_g", "_band", "_canopy"))
                and worst["max_rel_diff_vs_oracle"] <= 10.0 * tol and worst["max_rel_diff_vs_oracle"] <= 3.0 * sens):
            worst["explained_by_conditioning"] = ("worst element is a per-g-point diagnostic within 3x the oracle's own fma-vs-plain "
                                                  "movement there; broadband fluxes within tolerance")
    return worst


def check_parity_single(w, inputs):
    """Single precision (BASELINE configs[4]: the reference's PARKIND1_SINGLE build of the SPARTACUS solvers).  The reference's
    formulation is unstable there -- it warns (radiation_config.F90:1144) -- and which columns go wrong depends on the last bit,
    so neither the float build of the oracle nor the HIP path can be the yardstick for the other element by element.  Both are
    measured against the oracle in DOUBLE precision on every timed column: per field, the number of columns off by more
    than the tolerance and the median difference.  The HIP path passes if it is at least as close to double as the oracle's
    own float build is (columns off <= 1.5 x + 0.01 %, median <= 2 x + 1e-7)."""
    import copy
    tol = parity_tolerance(w.config)
    cfg_dp = copy.copy(w.config)
    cfg_dp.i_precision = 0
    osp = oracle_flux_of(w.config, inputs)
    odp = oracle_flux_of(cfg_dp, inputs)
    nchk = odp.ncol
    fields, ok = {}, True
    for name, t in w.case.flux_tensors.items():
        dp = odp.arrays.get(name)
        if dp is None:
            continue
        col_last = t.shape[-1] == w.ncol
        hip = (t[..., :nchk] if col_last else t[:nchk]).cpu().numpy()
        sp = osp.arrays[name]
        scale = np.maximum(np.abs(dp), 1e-3 * np.abs(dp).max() + 1e-300)
        ax = tuple(range(hip.ndim - 1)) if col_last else tuple(range(1, hip.ndim))

        def per_column(a):
            e = np.abs(a - dp) / scale
            e = np.where(np.isfinite(e), e, np.inf)
            return e.max(axis=ax) if e.ndim > 1 else e
        eh, eo = per_column(hip), per_column(sp)
        rec = {"columns_off_hip": int((eh > tol).sum()), "columns_off_oracle_float": int((eo > tol).sum()),
               "median_hip": float(np.median(eh)), "median_oracle_float": float(np.median(eo)),
               "nonfinite_columns_hip": int(np.isinf(eh).sum()), "nonfinite_columns_oracle_float": int(np.isinf(eo).sum())}
        # (a NaN / Inf column counts as "off" on either side; on top of that the GPU may not produce more of them than the
        #  oracle's float build -- the reference's own arithmetic -- does.  Since round 6 the float instantiation of the HIP shortwave
        #  solver keeps its albedo matrices within [0, 1] (kernel_spartacus.hip, section 4.1) and produces none on the synthetic workload)
        rec["ok"] = bool(rec["columns_off_hip"] <= 1.5 * rec["columns_off_oracle_float"] + 1.0e-4 * nchk + 1
                         and rec["median_hip"] <= 2.0 * rec["median_oracle_float"] + 1.0e-7
                         and rec["nonfinite_columns_hip"] <= rec["nonfinite_columns_oracle_float"])
        ok = ok and rec["ok"]
        fields[name] = rec
    worst = max(fields, key=lambda k: fields[k]["columns_off_hip"])
    nonfinite = {"columns_hip": max(v["nonfinite_columns_hip"] for v in fields.values()),
                 "columns_oracle_float": max(v["nonfinite_columns_oracle_float"] for v in fields.values())}
    summary = {k: fields[k] for k in ("sw_up", "sw_dn", "lw_up_clear", "lw_up", "lw_dn", "sw_dn_diffuse_surf_g") if k in fields}
    return {"reference": "oracle in double precision; yardstick: the oracle's own single-precision build (PARKIND1_SINGLE semantics)",
            "tolerance": tol, "columns_checked": int(nchk), "columns_timed": int(w.ncol), "ok": bool(ok),
            "fields_checked": len(fields), "fields_failed": [k for k, v in fields.items() if not v["ok"]], "nonfinite": nonfinite,
            "most_columns_off": {"field": worst, **fields[worst]}, "fields": summary}


def with_stdout_on_stderr(fn, *a, **kw):
    """(the reference's RRTMG set-up routines print to Fortran unit 6: keep stdout for the JSON line)"""
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        return fn(*a, **kw)
    finally:
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)


PCIE_GBS = None      # (h2d, d2h, both directions at once) of this box, measured once per process with page-locked buffers


def pcie_bandwidth(rad):
    global PCIE_GBS
    if PCIE_GBS is None:
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        if rad.lib.ecrad_hip_pcie_bandwidth(rad.handle, C.c_size_t(1 << 30), 3, C.byref(a), C.byref(b), C.byref(c)) != 0:
            raise RuntimeError(rad.lib.ecrad_hip_last_error(rad.handle).decode())
        PCIE_GBS = (a.value, b.value, c.value)
    return PCIE_GBS


def end_to_end_host(w, gpu_resident_value, repeats=3):
    """The same call through ECRAD_MEM_HOST pointers -- the mode every Fortran host uses: pageable host arrays, the call copies
    the column range in, runs the kernels and copies the results back, as a pipeline of column tiles on three streams
    (ecrad_amd/csrc/pipeline.hip: radiation_host_pipelined).  Next to it the ceiling the link sets: bytes per column in and out
    (ecrad_hip_last_call_info) over the measured host-to-device / device-to-host rates with both directions busy."""
    from ecrad_amd import abi
    from ecrad_amd.types import Flux
    ncol, nlev, sl, th, gas, cloud, aer = w.host_inputs
    if cloud is not None:
        frac0 = cloud.fraction.copy()
    flux = Flux.allocate(w.config, ncol, nlev)
    w.rad.radiation(ncol, nlev, 1, ncol, sl, th, gas, cloud, aer, flux)
    t = 0.0
    for _ in range(repeats):
        if cloud is not None:
            cloud.fraction[...] = frac0
        t0 = time.perf_counter()
        w.rad.radiation(ncol, nlev, 1, ncol, sl, th, gas, cloud, aer, flux)
        t += time.perf_counter() - t0
    info = abi.CallInfo()
    w.rad.lib.ecrad_hip_last_call_info(w.rad.handle, C.byref(info))
    progress("host-memory mode: link rates")
    h2d, d2h, duplex = pcie_bandwidth(w.rad)
    progress("host-memory mode: registered arrays")
    b_in, b_out = info.staged_in_bytes / ncol, info.staged_out_bytes / ncol
    # The ceiling: both directions at their one-way rates at once, i.e. the call can take no less than its larger transfer.
    # (`both_directions` -- two page-locked copies in opposite directions at once, one host thread each -- came out at the
    #  one-way rate on the boxes of round 4, less than what the pipelined call itself moves in both directions together:
    #  reported, not used.)
    ceiling = 1.0e9 / max(b_in / h2d, b_out / d2h)
    # ... and what the link sustains with BOTH directions busy (measured: two page-locked copies at once): the call moves b_in + b_out bytes
    # per column through it, so it cannot beat duplex / (b_in + b_out) either -- the tighter of the two ceilings whenever the duplex rate is
    # less than twice the one-way rate (round 6, /opt/rocm's runtime: 97 GB/s against 2 x 57)
    ceiling_duplex = 1.0e9 * duplex / (b_in + b_out)
    value = ncol * repeats / t
    # The same call on arrays the host keeps in page-locked memory (ecrad_hip_host_alloc: a host model's arrays live as long as it runs):
    # the copy engines read and write them directly instead of the runtime staging pageable memory through the calling threads.
    registered = None
    try:
        from ecrad_amd.interface import HostArrays, build_flux_struct, build_inputs_struct, relocate_call_arrays
        if cloud is not None:
            cloud.fraction[...] = frac0
        import copy
        arena = HostArrays(w.rad)
        # (shallow copies of the workload's objects: the page-locked arrays hang on the copies, the workload keeps its own)
        sl2, th2, gas2, cloud2, aer2 = (copy.copy(o) if o is not None else None for o in (sl, th, gas, cloud, aer))
        flux2 = Flux.allocate(w.config, ncol, nlev)
        try:
            moved = relocate_call_arrays(arena.copy_of, (sl2, th2, gas2, cloud2, aer2), flux2)
            cin, keep = build_inputs_struct(w.config, ncol, nlev, sl2, th2, gas2, cloud2, aer2)
            cflux = build_flux_struct(flux2)

            def call():
                if w.rad.lib.ecrad_hip_radiation(w.rad.handle, ncol, nlev, 1, ncol, C.byref(cin), C.byref(cflux)) != 0:
                    raise RuntimeError(w.rad.lib.ecrad_hip_last_error(w.rad.handle).decode())
            call()
            tr = 0.0
            for _ in range(repeats):
                if cloud2 is not None:
                    cloud2.fraction[...] = frac0
                t0 = time.perf_counter()
                call()
                tr += time.perf_counter() - t0
            same = all(np.array_equal(v, flux2.arrays[k], equal_nan=True) for k, v in flux.arrays.items())
            registered = {"value": ncol * repeats / tr, "unit": "columns/s", "ms_per_call": 1e3 * tr / repeats, "same_bits_as_pageable": bool(same),
                          "arrays_page_locked": len(moved), "bytes_page_locked": int(arena.nbytes), "how": "ecrad_hip_host_alloc"}
        finally:
            cin = cflux = keep = moved = sl2 = th2 = gas2 = cloud2 = aer2 = None      # (nothing may hold the memory that goes back now)
            flux2.arrays.clear()
            arena.close()
    except Exception as e:      # (additional information: never take the line down)
        registered = {"error": f"{type(e).__name__}: {e}"}
    return {"value": value, "unit": "columns/s", "ms_per_call": 1e3 * t / repeats, "registered_host_arrays": registered,
            "column_tiles": int(info.n_tiles), "tile_columns": int(info.tile_columns),
            "bytes_per_column": {"in": b_in, "out": b_out},
            "pcie_gbs": {"host_to_device": h2d, "device_to_host": d2h, "both_directions": duplex},
            "pcie_ceiling_columns_per_s": ceiling, "pcie_duplex_ceiling_columns_per_s": ceiling_duplex,
            "fraction_of_pcie_ceiling": value / min(ceiling, ceiling_duplex),
            "fraction_of_min_ceiling_and_value": value / min(ceiling, gpu_resident_value),
            "note": "ECRAD_MEM_HOST: pageable host arrays; copy-in, kernels and copy-out of consecutive column tiles overlap on three "
                    "streams (PCIe-inclusive); never `value`"}


class HostOnlyWorkload:
    """What end_to_end_host needs of a Workload, without a device batch and without torch: the configuration, the library, the columns on
    the host (the same seeded columns as the timed workload: ecrad_amd/synthetic.py)."""

    def __init__(self, name, ncol):
        from ecrad_amd.interface import Radiation
        from ecrad_amd.synthetic import make_columns
        self.name, self.ncol = name, ncol
        self.config, self.clear_sky, self.desc = build_config(name)
        self.rad = Radiation(self.config, backend="hip")
        self.host_inputs = make_columns(self.config, ncol, self.clear_sky)


def host_mode_child(name, ncol, gpu_resident_value, timeout=600):
    """end_to_end_host in a child process that never imports torch (ECRAD_BENCH_HOST_CHILD); returns its record or {"error": ...}."""
    import subprocess
    env = dict(os.environ, ECRAD_BENCH_HOST_CHILD=f"{name}:{ncol}:{gpu_resident_value!r}")
    env.pop("ECRAD_BENCH_WORKER", None)
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=timeout)
        if p.returncode != 0:
            return {"error": f"child exit {p.returncode}: {p.stderr[-300:]}"}
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}


def host_mode_child_main(spec):
    name, ncol, value = spec.split(":")
    assert "torch" not in sys.modules
    w = HostOnlyWorkload(name, int(ncol))
    r = end_to_end_host(w, float(value))
    assert "torch" not in sys.modules, "the host-only child must not load PyTorch's copy of the HIP runtime"
    r["hip_runtime"] = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln})
    r["process"] = "host-only child: libecrad_hip.so with /opt/rocm's HIP runtime, no PyTorch in the process (what a Fortran host loads)"
    w.rad.close()
    print(json.dumps(_finite(r), allow_nan=False), flush=True)


def small_blocks(name, nblock=80, nthreads=16, contexts=16, blocks_per_thread=32):
    """What an UNCHANGED blocked caller gets: `nthreads` host threads each calling radiation() on blocks of `nblock` columns
    of shared host arrays, as the reference's driver does with `!$OMP PARALLEL DO` over blocks (driver/ecrad_driver.F90:
    348-370; its test namelist has nblocksize = 80).  The calls run side by side on the contexts of the library's pool
    (include/ecrad_hip.h: ecrad_hip_set_concurrency); the same blocks from ONE thread are timed next to it."""
    import threading
    from ecrad_amd.interface import Radiation, build_flux_struct, build_inputs_struct
    from ecrad_amd.synthetic import make_columns
    from ecrad_amd.types import Flux
    config, clear_sky, _ = build_config(name)
    rad = Radiation(config, backend="hip", concurrency=(1, contexts))
    ncol = nblock * blocks_per_thread * nthreads
    n, nlev, sl, th, gas, cloud, aer = make_columns(config, ncol, clear_sky)
    flux = Flux.allocate(config, n, nlev)
    cin, keep = build_inputs_struct(config, n, nlev, sl, th, gas, cloud, aer)
    cflux = build_flux_struct(flux)
    out = {"columns_per_call": nblock, "contexts": contexts, "unit": "columns/s"}
    for nt in (nthreads, 1):
        blocks = [(i + 1, i + nblock) for i in range(0, nblock * blocks_per_thread * nt, nblock)]
        for _ in range(2):      # the first round is the warm-up: every context allocates its work arrays on its first call
            todo, lock, errors = list(blocks), threading.Lock(), []
            start = threading.Barrier(nt + 1)

            def worker():
                start.wait()
                while True:
                    with lock:
                        if not todo:
                            return
                        i0, i1 = todo.pop()
                    if rad.lib.ecrad_hip_radiation(rad.handle, n, nlev, i0, i1, C.byref(cin), C.byref(cflux)) != 0:
                        errors.append(rad.lib.ecrad_hip_last_error(rad.handle))
                        return
            threads = [threading.Thread(target=worker) for _ in range(nt)]
            for t in threads:
                t.start()
            rad.pool_info(reset=True)
            start.wait()
            t0 = time.perf_counter()
            for t in threads:
                t.join()
            dt = time.perf_counter() - t0
            if errors:
                raise RuntimeError(str(errors[0]))
        info = rad.pool_info()
        key = "value" if nt == nthreads else "one_thread"
        out[key] = len(blocks) * nblock / dt
        if nt == nthreads:
            out.update({"threads": nt, "max_calls_in_flight": info["max_in_flight"], "ms_per_call": 1e3 * dt * nt / len(blocks)})
    rad.close()
    del keep
    return out


def pool_mode(args):
    """`--threads-per-process T`: ONE process, no MPI, the library's pool of contexts over `--gpus N` devices, T host threads
    calling radiation() on blocks of `--block-columns` columns of host arrays -- the reference driver's own way of using a
    node (driver/ecrad_driver.F90:348-370), aimed at N x MI355X.  N x ncol columns per step (weak scaling); everything is
    host-memory mode, i.e. PCIe-inclusive: a different measurement from the GPU-resident `value` of the default mode, and
    labelled so."""
    import threading
    import torch
    from ecrad_amd.interface import Radiation, build_flux_struct, build_inputs_struct
    from ecrad_amd.synthetic import make_columns
    from ecrad_amd.types import Flux
    ndev = args.gpus
    # (ECRAD_HIP_FAKE_DEVICES=N, a TEST switch of the library: N device slots all mapped onto this box's GPU -- the dry run of
    #  an N-GPU node's pool on a 1-GPU box, tests/test_bench_launcher.py; the line says so and its numbers mean nothing)
    fake = int(os.environ.get("ECRAD_HIP_FAKE_DEVICES", "0") or 0)
    if max(torch.cuda.device_count(), fake) < ndev:
        print(f"bench.py: --gpus {ndev} asked for, {torch.cuda.device_count()} GPU(s) visible on this node", file=sys.stderr)
        sys.exit(2)
    config, clear_sky, desc = build_config(args.workload)
    nthreads = args.threads_per_process
    rad = Radiation(config, backend="hip", concurrency=(ndev, max(2, -(-nthreads // ndev))))
    base = make_columns(config, args.ncol, clear_sky)
    nlev = base[1]
    ncol = args.ncol * ndev
    # the batch of every device is the same `ncol` synthetic columns: tiled along the column axis
    objs = []
    for obj in base[2:]:
        if obj is None:
            objs.append(None)
            continue
        import copy
        o = copy.copy(obj)
        for k, v in list(vars(o).items()):
            if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[-1] == args.ncol:
                setattr(o, k, np.ascontiguousarray(np.concatenate([v] * ndev, axis=-1)))
        objs.append(o)
    sl, th, gas, cloud, aer = objs
    frac0 = None if cloud is None else cloud.fraction.copy()
    flux = Flux.allocate(config, ncol, nlev)
    cin, keep = build_inputs_struct(config, ncol, nlev, sl, th, gas, cloud, aer)
    cflux = build_flux_struct(flux)
    nb = args.block_columns
    blocks = [(i + 1, min(ncol, i + nb)) for i in range(0, ncol, nb)]

    def step():
        if frac0 is not None:
            cloud.fraction[...] = frac0
        todo, lock, errors = list(blocks), threading.Lock(), []

        def worker():
            while True:
                with lock:
                    if not todo:
                        return
                    i0, i1 = todo.pop()
                if rad.lib.ecrad_hip_radiation(rad.handle, ncol, nlev, i0, i1, C.byref(cin), C.byref(cflux)) != 0:
                    errors.append(rad.lib.ecrad_hip_last_error(rad.handle))
                    return
        threads = [threading.Thread(target=worker) for _ in range(nthreads)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise RuntimeError(str(errors[0]))
    for _ in range(args.warmup):
        step()
    rad.pool_info(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    info = rad.pool_info()
    # the first `ncol` columns are what device-resident runs produce: the same fluxes, whichever device ran a block
    same = all(np.array_equal(a[..., :args.ncol], a[..., -args.ncol:]) if a.shape[-1] == ncol else np.array_equal(a[:args.ncol], a[-args.ncol:])
               for a in flux.arrays.values()) if ndev > 1 else True
    out = {"metric": "columns/sec (SW+LW) at 137 lev, " + ("RRTMG 140/112" if desc["rrtmg"] else f"ecCKD-{config.n_g_sw}"),
           "value": ncol * args.steps / elapsed, "unit": "columns/s", "n_gpus": ndev, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "mode": "host_memory_pool: one process, the library's pool of contexts over the devices, host arrays in and out of every call "
                   "(PCIe-inclusive) -- not the GPU-resident `value` of the default mode",
           "config": {"workload": args.workload, "columns_per_gpu_per_step": args.ncol, "nlev": nlev, "n_g_sw": config.n_g_sw, "n_g_lw": config.n_g_lw,
                      "sw_solver": desc["sw_solver"], "threads_per_process": nthreads, "block_columns": nb,
                      "parallelism": f"blocks of columns spread over {ndev} GPU(s) by {nthreads} host threads of one process, no collective"},
           "pool": info, "blocks_identical_across_devices": bool(same)}
    if fake:
        out["test_fake_devices"] = f"{fake} device slots mapped onto one GPU (ECRAD_HIP_FAKE_DEVICES): exercises the pool's code, not a measurement"
    rad.close()
    del keep
    emit(out)
    sys.exit(0 if same else 1)


def measure(name, ncol, steps, warmup, rank, local_rank, world, barrier, allreduce_max, do_cpu, do_host_mode, regions=1):
    """`regions` > 1 (the extra workloads, five steps each): the K-step region is timed that many times (three) and the MEDIAN is the
    workload's figure; every region's ms per step is in the record (`timed_regions`), so that one hiccup of the box (35 ms once, seen in
    round 5: 7 ms per step of a five-step region) is visible and neither kept nor picked against.  The headline is timed ONCE, K steps exactly."""
    sample_cols = 16384 if do_cpu else 0
    progress(f"{name} ncol={ncol}: set-up")
    w = Workload(name, ncol, rank, local_rank, sample_cols)
    progress(f"{name}: timed steps")
    elapsed_rank = timed_steps(w, steps, warmup, barrier)
    region_ms = [1e3 * elapsed_rank / steps]
    region_s = [elapsed_rank]
    for _ in range(regions - 1):
        e = timed_steps(w, steps, 0, barrier)
        region_ms.append(1e3 * e / steps)
        region_s.append(e)
    elapsed_rank = sorted(region_s)[(len(region_s) - 1) // 2]      # the median (of three); the lower middle of an even count
    elapsed = allreduce_max(elapsed_rank)
    elapsed_min = -allreduce_max(-elapsed_rank)
    stage = w.stage_ms()
    # a few extra untimed steps to average the per-stage duration
    import torch
    acc = [stage]
    for _ in range(2):
        w.step()
        torch.cuda.synchronize()
        acc.append(w.stage_ms())
    stage = {k: float(np.mean([a[k] for a in acc])) for k in stage}
    res = {"value": world * ncol * steps / elapsed, "unit": "columns/s", "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * elapsed / steps, "dtype": "f32" if getattr(w.config, "i_precision", 0) == 1 else "f64",
           "config": {"workload": name, "columns_per_gpu_per_step": ncol, "nlev": w.nlev, "n_g_sw": w.config.n_g_sw,
                      "n_g_lw": w.config.n_g_lw, "sw_solver": w.desc["sw_solver"], "gas_model": "RRTMG-IFS" if w.desc["rrtmg"] else "ecCKD",
                      "aerosols": bool(w.config.use_aerosols), "clouds": not w.clear_sky},
           "roofline": roofline_of(w, stage, elapsed / steps)}
    if regions > 1:
        res["timed_regions"] = {"count": regions, "ms_per_step_each": region_ms, "kept": "median", "fastest": min(region_ms), "slowest": max(region_ms)}
    if world > 1:
        # every rank times the same K steps between the same two barriers: the slowest rank is `ms_per_step`
        res["ms_per_step_ranks"] = {"min": 1e3 * elapsed_min / steps, "max": 1e3 * elapsed / steps}
        e2 = allreduce_max(timed_steps(w, steps, 1, barrier, gather_world=world))
        res["value_with_gather"] = world * ncol * steps / e2
        res["ms_per_step_with_gather"] = 1e3 * e2 / steps
        if rank == 0:      # rank 0 holds every rank's profiles: (ranks, fields, half levels, columns per rank)
            res["gathered"] = {"ranks": len(w.gathered), "shape_per_rank": list(w.gathered[0].shape), "fields": w.profile_names,
                               "bytes_received": int(sum(b.numel() * b.element_size() for b in w.gathered[1:]))}
        # (TEST HOOK: ECRAD_BENCH_TEST_LIBRARY_GATHER_ANYWAY=1 runs the leg although the ranks share a GPU -- RCCL then refuses the communicator,
        #  which is how tests/test_bench_launcher.py exercises what happens to the line when the leg fails on every rank)
        if ((os.environ.get("ECRAD_BENCH_TEST_SHARED_GPU") != "1" or os.environ.get("ECRAD_BENCH_TEST_LIBRARY_GATHER_ANYWAY") == "1")
                and os.environ.get("ECRAD_BENCH_NO_LIBRARY_GATHER") != "1"):
            progress(f"{name}: library gather")
            res["library_gather"] = library_gather_leg(w, world, rank, steps, barrier, allreduce_max)
            if LIBRARY_GATHER_HUNG:
                return res      # (the stuck thread holds the context: closing the workload would wait for it)
    if world == 1 and os.environ.get("ECRAD_BENCH_FORCE_LIBRARY_GATHER") == "1":      # (tests: the leg on a one-rank communicator)
        res["library_gather"] = library_gather_leg(w, 1, 0, steps, barrier, allreduce_max)
    if do_cpu and rank == 0:
        progress(f"{name}: cpu baseline")
        w.step()
        torch.cuda.synchronize()
        res["cpu_baseline"], oracle_flux = with_stdout_on_stderr(cpu_baseline, w.config, w.sample, workload_name=name)
        inputs_all = w.sample
        if w.host_inputs is not None and w.host_inputs[0] > w.sample[0]:
            # every timed column (batches of up to CHUNK_COLUMNS columns are still on the host), not only the timing sample
            inputs_all = w.host_inputs
            if getattr(w.config, "i_precision", 0) != 1:
                oracle_flux = with_stdout_on_stderr(oracle_flux_of, w.config, inputs_all)
        progress(f"{name}: parity")
        if getattr(w.config, "i_precision", 0) == 1:
            res["parity"] = with_stdout_on_stderr(check_parity_single, w, inputs_all)
        else:
            res["parity"] = with_stdout_on_stderr(check_parity, w, oracle_flux, inputs_all)
    if do_host_mode and rank == 0 and w.host_inputs is not None:
        progress(f"{name}: host-memory mode")
        # In THIS process the HIP runtime is the copy PyTorch bundles (it is imported first: tests/conftest.py), under which two page-locked
        # copies in opposite directions share ONE direction's rate (57 GB/s together on the round-6 boxes) and pageable copies stage slowly.
        # A Fortran host has /opt/rocm's runtime: both directions at once reach 97 GB/s there (profiles/NOTES_r06.md section 3).  So the
        # figure of record comes from a child process that never loads torch -- the library as its real callers load it -- and the
        # in-process figure stays next to it.
        here = end_to_end_host(w, res["value"] / world)
        child = host_mode_child(name, w.ncol, res["value"] / world)
        if child is not None and "value" in child:
            child["under_pytorch_runtime"] = {k: here.get(k) for k in ("value", "ms_per_call", "pcie_gbs", "pcie_ceiling_columns_per_s")}
            child["under_pytorch_runtime"]["page_locked_value"] = (here.get("registered_host_arrays") or {}).get("value")
            res["end_to_end_host"] = child
        else:
            here["host_only_child"] = child
            res["end_to_end_host"] = here
    progress(f"{name}: close")
    w.close()
    progress(f"{name}: done")
    return res


LINE_LIMIT = 4096          # the driver keeps a bounded tail of stdout: the final line has to fit it with room to spare


def _finite(x):
    """JSON has no NaN / Infinity: non-finite floats become null, numpy scalars become Python ones (recursively)."""
    if isinstance(x, dict):
        return {str(k): _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, (np.floating, np.integer, np.bool_)):
        x = x.item()
    if isinstance(x, float) and not np.isfinite(x):
        return None
    return x


def _sig(x, n=6):
    """Floats of the compact line carry n significant digits (the detail file keeps every digit)."""
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, list):
        return [_sig(v, n) for v in x]
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    return x


def _compact_roofline(r):
    keep = ("bound", "binding_roof", "fractions", "valu", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
            "kernel_ms", "stage_ms", "algorithmic_bytes_per_column", "launches_per_step", "scratch_mantissa_bits", "measured_triad", "measured_read", "measured_copy")
    out = {k: r[k] for k in keep if k in r}
    if "whole_step" in r:
        out["whole_step_frac"] = r["whole_step"]["frac"]
    if "compute" in r:
        out["compute"] = {k: r["compute"][k] for k in ("bound", "achieved", "peak", "unit", "frac")}
    return out


def _compact_cpu(c):
    out = {k: c[k] for k in ("value", "unit", "cores", "kind") if k in c}
    sample = c.get("sample", "")
    out["sample"] = sample if len(sample) <= 200 else sample[:197] + "..."
    port = c.get("components", {}).get("port")
    if port and "value" in port:
        out["port_value"] = port["value"]       # the oracle's own figure next to the reference executable's
    return out


def _compact_parity(p):
    keep = ("ok", "max_rel_diff_vs_oracle", "max_rel_diff_broadband", "field", "columns_checked", "tolerance",
            "oracle_fma_vs_plain_there", "fields_checked", "fields_failed", "nonfinite")
    return {k: p[k] for k in keep if k in p}


def compact_line(out, detail_path=None):
    """The ONE line the driver parses: the contract's keys + `roofline` + `cpu_baseline` + a parity summary + one short
    record per extra workload.  Everything else of `out` (per-field single-precision parity, cpu_baseline.components,
    small_blocks, end_to_end_host, per-stage times) is in the detail file (gpurun_out/bench_detail.json).  Strict JSON: non-finite
    floats are null (`allow_nan=False`).  Always shorter than LINE_LIMIT (tests/test_bench_line.py)."""
    out = _finite(out)
    line = {k: out[k] for k in ("metric", "value", "value_exact_scratch", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "ms_per_step_exact_scratch", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data", "config") if k in out}
    for k in ("ms_per_step_ranks", "value_with_gather", "ms_per_step_with_gather", "gathered", "library_gather", "mode", "test_shared_gpu",
              "test_fake_devices", "blocks_identical_across_devices", "pool", "attempts", "aborted_attempts", "fault"):
        if k in out:
            line[k] = out[k]
    if "roofline" in out:
        line["roofline"] = _compact_roofline(out["roofline"])
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _compact_cpu(out["cpu_baseline"])
    if "parity" in out:
        line["parity"] = _compact_parity(out["parity"])
    if "end_to_end_host" in out:
        e = out["end_to_end_host"]
        line["end_to_end_host"] = {k: e[k] for k in ("value", "unit", "pcie_ceiling_columns_per_s", "pcie_duplex_ceiling_columns_per_s") if k in e}
        if "under_pytorch_runtime" in e:
            line["end_to_end_host"]["runtime"] = "/opt/rocm (host-only child)"
            line["end_to_end_host"]["value_under_pytorch_runtime"] = e["under_pytorch_runtime"].get("value")
        if isinstance(e.get("registered_host_arrays"), dict) and "value" in e["registered_host_arrays"]:
            line["end_to_end_host"]["registered_value"] = e["registered_host_arrays"]["value"]
    if "workloads" in out:
        wl = {}
        for name, r in out["workloads"].items():
            if "error" in r:
                wl[name] = {"error": str(r["error"])[:120]}
                continue
            rec = {"value": r.get("value"), "ms_per_step": r.get("ms_per_step"), "ncol": r.get("config", {}).get("columns_per_gpu_per_step")}
            if "roofline" in r:
                rec["frac"] = r["roofline"]["frac"]
                rec["kernel_ms"] = r["roofline"]["kernel_ms"]
            if "parity" in r:
                rec["parity_ok"] = r["parity"]["ok"]
                if "nonfinite" in r["parity"]:      # single precision: columns with a non-finite flux, HIP / the oracle's float build
                    rec["nonfinite_columns"] = {"hip": r["parity"]["nonfinite"]["columns_hip"], "oracle_float": r["parity"]["nonfinite"]["columns_oracle_float"]}
            if "cpu_baseline" in r:
                rec["cpu"] = r["cpu_baseline"].get("value")
            wl[name] = rec
        line["workloads"] = wl
    if "small_blocks" in out:
        line["small_blocks"] = {k: (v.get("value") if "error" not in v else None) for k, v in out["small_blocks"].items()}
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(_sig(line), allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:      # (cannot happen with the workloads of this file; a longer list sheds its optional parts)
        for k in ("small_blocks", "workloads", "end_to_end_host"):
            line.pop(k, None)
            text = json.dumps(_sig(line), allow_nan=False, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    return text


def emit(out):
    """Write the full record to the detail file, print the compact line LAST on standard output."""
    full = _finite(out)
    detail_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT
    path = os.path.join(detail_dir, "bench_detail.json")
    rel = None
    try:
        with open(path, "w") as f:
            json.dump(full, f, allow_nan=False, indent=1)
        rel = os.path.relpath(path, ROOT)
    except OSError:
        pass
    # (NOT echoed on standard error: the driver's record is a bounded tail of stdout AND stderr together)
    print(f"bench.py: full record in {rel}" if rel else "bench.py: the detail file could not be written", file=sys.stderr)
    sys.stderr.flush()
    sys.stdout.flush()
    print(compact_line(out, rel), flush=True)


def supervise():
    """The single-GPU run as a watched child process, so that a run the ROCm runtime ends from outside Python (SIGABRT after "Memory access
    fault by GPU", round 5: profiles/NOTES_r05.md section 15, root cause and fix profiles/NOTES_r06.md section 1) still leaves a record: the
    parent prints ONE line `{"metric": ..., "value": null, "fault": true, "signal": ..., "last_phase": ...}` and exits with 128 + signal.
    A run is NOT started again: a crash of the library is a failed bench.  ECRAD_BENCH_RETRY=N (a debugging aid, never set by bench.py
    itself or by the tests of the default path) allows N further attempts after SIGABRT; the line of an attempt that then completes carries
    `"fault": true`, `attempts` and the aborted attempts, and the exit status is 70, not 0.  Not under torchrun (WORLD_SIZE > 1).
    ECRAD_BENCH_NO_SUPERVISOR=1 runs the measurement in this process."""
    import signal
    import subprocess
    aborted = []
    retries = max(0, int(os.environ.get("ECRAD_BENCH_RETRY", "0")))
    for attempt in range(1, retries + 2):
        env = dict(os.environ, ECRAD_BENCH_WORKER="1", ECRAD_BENCH_ATTEMPT=str(attempt), ECRAD_BENCH_ABORTED=json.dumps(aborted))
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env)
        try:
            rc = child.wait()
        except KeyboardInterrupt:
            child.kill()
            raise
        if rc >= 0:
            sys.exit(70 if (rc == 0 and aborted) else rc)
        last = None
        try:
            with open(os.path.join(ROOT, "gpurun_out", "bench_progress.log")) as f:
                mine = [ln.split(None, 3)[3].strip() for ln in f if ln.split(None, 3)[1:3] == ["pid", str(child.pid)]]
            last = mine[-1] if mine else None
        except (OSError, IndexError):
            pass
        aborted.append({"attempt": attempt, "signal": -rc, "last_phase": last})
        print(f"bench.py: attempt {attempt} ended by signal {-rc} in phase {last!r}", file=sys.stderr)
        if -rc != signal.SIGABRT:
            break
    sys.stderr.flush()
    print(json.dumps({"metric": "columns/sec (SW+LW) at 137 lev, ecCKD-32; 1/2/4/8 GPU + %HBM roofline", "value": None, "unit": "columns/s",
                      "n_gpus": 1, "higher_is_better": True, "fault": True, "signal": aborted[-1]["signal"],
                      "last_phase": aborted[-1]["last_phase"], "aborted_attempts": aborted}), flush=True)
    sys.exit(128 + aborted[-1]["signal"])


def main():
    os.environ.setdefault("GFORTRAN_UNBUFFERED_ALL", "1")
    if os.environ.get("ECRAD_BENCH_HOST_CHILD"):
        return host_mode_child_main(os.environ["ECRAD_BENCH_HOST_CHILD"])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ncol", type=int, default=100000, help="columns per GPU per step of the headline workload")
    ap.add_argument("--workload", default="clear_homogeneous_ecckd32", help="headline workload (ecrad_amd/synthetic.py: BENCH_CONFIGS)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the other BASELINE configurations (\"workloads\")")
    ap.add_argument("--no-host-mode", action="store_true",
                    help="skip end_to_end_host / small_blocks (the profiling runs: their column tiles and small batches are launches of the "
                         "same kernels and would enter the per-kernel averages)")
    ap.add_argument("--threads-per-process", type=int, default=0,
                    help="> 0: ONE process whose host threads call radiation() on blocks of host arrays, spread over --gpus devices by the "
                         "library's pool of contexts (host-memory mode, PCIe-inclusive: see pool_mode)")
    ap.add_argument("--block-columns", type=int, default=12500, help="columns per call in --threads-per-process mode")
    args = ap.parse_args()
    if (args.gpus == 1 and args.threads_per_process == 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1
            and os.environ.get("ECRAD_BENCH_WORKER") != "1" and os.environ.get("ECRAD_BENCH_NO_SUPERVISOR") != "1"):
        return supervise()
    if os.environ.get("ECRAD_BENCH_WORKER") == "1":
        try:      # (a worker does not outlive the process that watches it)
            C.CDLL(None).prctl(1, 9)
        except (OSError, AttributeError):
            pass
        # TEST HOOK (tests/test_bench_line.py): the first N attempts end the way a GPU memory fault ends a process
        if int(os.environ.get("ECRAD_BENCH_ATTEMPT", "1")) <= int(os.environ.get("ECRAD_BENCH_TEST_ABORT_ATTEMPTS", "0")):
            progress("test hook: abort")
            os.abort()
    if args.threads_per_process > 0:
        return pool_mode(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # TEST MODE (tests/test_bench_launcher.py on a 1-GPU box): the N ranks share the visible GPU(s), rendezvous over gloo
    # and stage the gather through the host -- it exercises the N>1 code of this file, its numbers mean nothing.
    shared = os.environ.get("ECRAD_BENCH_TEST_SHARED_GPU") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under torch.distributed.run: start the N ranks ourselves (one process per GPU, same argv) and wait for them
        have = torch.cuda.device_count()
        if have < args.gpus and not shared:
            print(f"bench.py: --gpus {args.gpus} asked for, {have} GPU(s) visible on this node", file=sys.stderr)
            sys.exit(2)
        from ecrad_amd.parallel import launch_ranks
        sys.exit(launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if shared:
        local_rank = local_rank % torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local_rank}, {torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo" if shared else "nccl", rank=rank, world_size=world)
        one = torch.ones(1, dtype=torch.float64, device="cpu" if shared else f"cuda:{local_rank}")
        dist.all_reduce(one)                      # an actual RCCL collective over all ranks: its sum IS the rank count
        rccl_ranks = int(round(float(one.item())))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allreduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if shared else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    do_cpu = world == 1 and not args.no_cpu_baseline
    head = measure(args.workload, args.ncol, args.steps, args.warmup, rank, local_rank, world, barrier, allreduce_max,
                   do_cpu, do_host_mode=(world == 1 and not args.no_host_mode))
    cfg = head["config"]
    out = {
        "metric": "columns/sec (SW+LW) at 137 lev, " + ("RRTMG 140/112" if cfg["gas_model"] != "ecCKD" else f"ecCKD-{cfg['n_g_sw']}"),
        "value": head["value"], "unit": "columns/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": head.get("dtype", "f64"), "data": "synthetic",
        "config": dict(cfg, parallelism=f"columns sharded over {world} GPU(s), no data-path collective"),
        "roofline": head["roofline"],
    }
    if shared:
        out["test_shared_gpu"] = "ranks share the visible GPU(s) over gloo: exercises the N>1 code only, not a measurement"
    for k in ("ms_per_step_ranks", "value_with_gather", "ms_per_step_with_gather", "gathered", "library_gather", "cpu_baseline", "parity", "end_to_end_host"):
        if k in head:
            out[k] = head[k]
    failed = "parity" in head and not head["parity"]["ok"]
    if (world == 1 and head["roofline"].get("scratch_mantissa_bits") == 39 and os.environ.get("ECRAD_HIP_EXACT_SCRATCH") != "1"
            and not args.no_host_mode):      # (--no-host-mode is what the profiling runs pass: their per-kernel tables hold the shipped kernels only)
        # the same K steps with whole doubles in the shortwave sweep records (the library's ECRAD_HIP_EXACT_SCRATCH=1: the instantiations of
        # kernel_ica_sw_exact.hip / kernel_tc_sw_exact.hip, 40 bytes per record instead of 32): what the 39-bit packing buys, next to `value`
        try:
            progress("exact-scratch leg")
            os.environ["ECRAD_HIP_EXACT_SCRATCH"] = "1"
            try:
                wx = Workload(args.workload, args.ncol, rank, local_rank, 0)
            finally:
                del os.environ["ECRAD_HIP_EXACT_SCRATCH"]
            ex = timed_steps(wx, args.steps, args.warmup, barrier)
            out["value_exact_scratch"] = args.ncol * args.steps / ex
            out["ms_per_step_exact_scratch"] = 1e3 * ex / args.steps
            wx.close()
        except Exception as e:
            out["value_exact_scratch"] = None
            out["exact_scratch_error"] = f"{type(e).__name__}: {e}"
    if world == 1 and not args.headline_only and args.workload == "clear_homogeneous_ecckd32":
        out["workloads"] = {}
        for name, ncol in EXTRA_WORKLOADS:
            steps = max(2, min(args.steps, 5 if ncol <= 100000 else 3))
            try:
                r = measure(name, ncol, steps, 1, rank, local_rank, world, barrier, allreduce_max, do_cpu,
                            do_host_mode=(name == "tripleclouds_ecckd32" and ncol <= CHUNK_COLUMNS and not args.no_host_mode), regions=3)
            except Exception as e:      # an extra workload must not take the headline line down with it
                r = {"error": f"{type(e).__name__}: {e}"}
            if "parity" in r and not r["parity"]["ok"]:
                r["value"] = None
                failed = True
            out["workloads"][name if name not in out["workloads"] else f"{name}_{ncol}"] = r
    if world == 1 and not args.headline_only and not args.no_host_mode and args.workload == "clear_homogeneous_ecckd32":
        # the boundary as an unchanged blocked caller uses it (blocks of 80 columns, 16 host threads, host arrays)
        out["small_blocks"] = {}
        for name in ("clear_homogeneous_ecckd32", "tripleclouds_ecckd32"):
            try:
                progress(f"small blocks {name}")
                out["small_blocks"][name] = small_blocks(name)
            except Exception as e:
                out["small_blocks"][name] = {"error": f"{type(e).__name__}: {e}"}
    if os.environ.get("ECRAD_BENCH_WORKER") == "1":
        out["attempts"] = int(os.environ.get("ECRAD_BENCH_ATTEMPT", "1"))
        prior = json.loads(os.environ.get("ECRAD_BENCH_ABORTED", "[]"))
        if prior:
            out["aborted_attempts"] = prior
            out["fault"] = True
    progress("emit")
    if rank == 0:
        if failed and "parity" in head and not head["parity"]["ok"]:
            out["value"] = None
        emit(out)
    if LIBRARY_GATHER_HUNG or LIBRARY_GATHER_FAILED:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1 if failed else 0)      # (a thread of this or another rank is stuck in RCCL: no barrier, no clean-up)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
