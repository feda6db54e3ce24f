#!/usr/bin/env python
"""tools/dropin_blocks.py [ncol] -- what the reference's OWN driver gets from the drop-in when it is run the way the IFS runs radiation: blocks of 80
columns (nblocksize of test/ifs/configCY49R1_ecckd.nam) dealt to OpenMP threads (driver/ecrad_driver.F90:348-370), every thread calling
radiation() -> ecrad_hip_radiation on host arrays.  The driver compiled WITH OpenMP around the drop-in (tests/_build/dropin_omp/ecrad_hip), its own
timer ("Time elapsed in radiative transfer"), OMP_NUM_THREADS = 16 ... 128, and the library's pool report.  Synthetic columns of two workloads."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from ecrad_amd.driver import save_inputs
from ecrad_amd.synthetic import make_columns
from test_fortran_dropin import OMP_EXE, _pool_report, write_namelist

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
for workload, edits in (("clear_homogeneous_ecckd32", {"sw_solver_name": '"Homogeneous"', "lw_solver_name": '"Homogeneous"', "use_aerosols": "false"}),
                        ("tripleclouds_ecckd32", {})):
    config, clear_sky, _ = bench.build_config(workload)
    inputs = make_columns(config, ncol, clear_sky)
    with tempfile.TemporaryDirectory() as tmp:
        inp = os.path.join(tmp, "inputs.nc")
        save_inputs(inp, config, *inputs[2:])
        write_namelist(os.path.join(tmp, "base.nam"), dict({"do_save_spectral_flux": "false", "iverbose": "1", "iverbosesetup": "0"}, **edits))
        base = open(os.path.join(tmp, "base.nam")).read()
        for nblock, threads in ([(80, int(os.environ.get("TRACE_THREADS", "16")))] if (os.environ.get("ECRAD_HIP_BATCH_TRACE") or os.environ.get("TRACE_THREADS")) else [(80, 16), (80, 64), (320, 16), (1280, 16), (5120, 8), (ncol, 1)]):
            nam, out = os.path.join(tmp, "c.nam"), os.path.join(tmp, "out.nc")
            open(nam, "w").write(re.sub(r"nrepeat\s*=\s*\d+", "nrepeat = 20", re.sub(r"nblocksize\s*=\s*\d+", f"nblocksize = {nblock}", base)))
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_STACKSIZE="1G", ECRAD_HIP_CONTEXTS="16", ECRAD_HIP_DEVICES="1", ECRAD_HIP_POOL_REPORT="1")
            p = subprocess.run(f"ulimit -s unlimited; exec {OMP_EXE} {nam} {inp} {out}", shell=True, capture_output=True, text=True, cwd=tmp, env=env, timeout=900)
            text = p.stdout + p.stderr
            m = re.search(r"Time elapsed in radiative transfer:\s*([0-9.Ee+-]+)\s*seconds", text)
            if p.returncode != 0 or not m:
                print(workload, nblock, threads, "FAILED", text[-400:]); continue
            t = float(m.group(1)) / 20.0
            if os.environ.get("ECRAD_HIP_BATCH_TRACE"):      # (the library's own phase times of every batch: the last few)
                print("\n".join([ln[:330] for ln in text.splitlines() if ln.startswith("ecrad_hip batch")][-10:]))
            pool = _pool_report(text)
            print(f"{workload}: {ncol} columns, blocks of {nblock}, {threads} OpenMP threads: {t*1e3:.1f} ms per pass -> {ncol/t:.0f} columns/s; "
                  f"calls {pool['calls']}, max in flight {pool['max_in_flight']}, batches {pool['batches']}", flush=True)
