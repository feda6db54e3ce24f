"""The drop-in executed inside its host (BOUNDARY PROOF, not an oracle pin).

tests/_build/dropin/ecrad_hip (tools/build_dropin.py, built where /root/reference exists; the binary travels) is the
reference's OWN offline driver -- driver/ecrad_driver.F90 with its namelist reader, input reader, setup_radiation table
preparation, save_fluxes, every radiation/ module -- compiled unmodified from where it lies, EXCEPT that the module
radiation_interface is ecrad_amd/fortran/radiation_interface.F90 (radiation() -> radiation_hip() over the reference's
real derived types, -DECRAD_HIP_REFERENCE_TYPES) and the netCDF library is ecrad_amd/fortran/netcdf.F90 + nc_classic.c.
Here it runs the reference's test/ifs configurations (test/ifs/Makefile targets, namelists made from
configCY49R1_ecckd.nam by change_namelist-style edits) on test/ifs/ecrad_meridian.nc and the output FILE must equal the
reference's golden output file of that target, variable by variable, to float32 rounding."""
import os
import re
import subprocess

import numpy as np
import pytest

from ecrad_amd.ncfile import NcFile
from helpers import GOLDEN_DIR, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "dropin", "ecrad_hip")
NAMELIST = os.path.join(GOLDEN_DIR, "configCY49R1_ecckd.nam")
MERIDIAN = os.path.join(GOLDEN_DIR, "ecrad_meridian.nc")
DATA_DIR = os.path.join(ROOT, "data")
FLOAT32_TOL = 2.0e-7

# test/ifs/configCY49R1.nam as its differences from configCY49R1_ecckd.nam (diff of the two files)
RRTMG = {"use_general_cloud_optics": "false", "liquid_model_name": '"SOCRATES"', "ice_model_name": '"Fu-IFS"',
         "sw_solver_name": '"McICA"', "lw_solver_name": '"McICA"', "gas_model_name": '"RRTMG-IFS"',
         "do_surface_sw_spectral_flux": "true", "do_cloud_aerosol_per_sw_g_point": "false",
         "do_cloud_aerosol_per_lw_g_point": "false", "do_nearest_spectral_lw_emiss": "true", "do_weighted_surface_mapping": "false"}
# the targets of test/ifs/Makefile that have a golden file (:34-118): namelist family + change_namelist.sh arguments
TARGETS = {
    "ecckd_mcica": ({}, {"sw_solver_name": '"McICA"', "lw_solver_name": '"McICA"'}),
    "default": (RRTMG, {}),
    "noaer": (RRTMG, {"use_aerosols": "false"}),
    "expexp": (RRTMG, {"overlap_scheme_name": '"Exp-Exp"'}),
    "tripleclouds": (RRTMG, {"sw_solver_name": '"Tripleclouds"', "lw_solver_name": '"Tripleclouds"'}),
    "cloudless": (RRTMG, {"use_aerosols": "false", "sw_solver_name": '"Cloudless"', "lw_solver_name": '"Cloudless"'}),
}


def write_namelist(path, *edits):
    """change_namelist.sh: set key = value in the &radiation group (replace the assignment, or add it)."""
    text = open(NAMELIST).read()
    changes = {"directory_name": f'"{DATA_DIR}"'}
    for e in edits:
        changes.update(e)
    head, rad = text.split("&radiation\n", 1)
    for k, v in changes.items():
        pat = re.compile(r"^(\s*)" + re.escape(k) + r"\s*=[^,\n]*,?", re.M)
        if pat.search(rad):
            rad = pat.sub(lambda m: f"{m.group(1)}{k} = {v},", rad, count=1)
        else:
            rad = f"{k} = {v},\n" + rad
    with open(path, "w") as f:
        f.write(head + "&radiation\n" + rad)



# What a GPU process that could not START prints (the HIP runtime found no usable device, or could not initialise): the one
# kind of failure that is the box's and not the product's.  Round 3 saw one such failure in six full runs of this file and
# retried on ANY non-zero return code; round 4 ran the 54 GPU cases three times over without any retry (162 processes,
# gpurun_out/r04_b) and saw none.  The retry is now limited to this signature, before any kernel ran, every retry is
# written to gpurun_out/dropin_retries.log, and the last test of this file fails if it was needed more than twice.
LW_SP_ENVELOPE = 4.0      # all-sky longwave of single-precision SPARTACUS: HIP sp vs dp may be this many times the reference sp vs dp (round 6: 0.0029 / 0.0031 and 0.0355 / 0.0354; the reference's own figure moves with the host's math library, hence the margin and the 5 % floor)
STARTUP_SIGNATURES = ("no usable MI355X device", "hipErrorNoDevice", "no ROCm-capable device", "hipErrorInvalidDevice",
                      "hipErrorNotInitialized", "hipErrorInitializationError", "Unable to open /dev/kfd", "HSA_STATUS_ERROR_OUT_OF_RESOURCES")
RETRIES = []


def _run(*args, **kw):
    """subprocess.run of one of the executables.  A process that failed to START on the GPU (STARTUP_SIGNATURES in its
    output, and nothing of the radiation calculation) is run once more and the retry is recorded; any other failure --
    wrong results, a crash or a hang of the code under test -- is returned as it is."""
    import sys
    import time
    p = subprocess.run(*args, **kw)
    if p.returncode != 0 and not os.environ.get("ECRAD_TEST_NO_RETRY"):
        text = (p.stdout or "") + (p.stderr or "")
        started = "Time elapsed in radiative transfer" in text or "Writing" in text
        if any(sig in text for sig in STARTUP_SIGNATURES) and not started:
            RETRIES.append(text[-1500:])
            sys.stderr.write("GPU process failed to start (return code %d), retrying once:\n%s\n" % (p.returncode, text[-1500:]))
            try:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "dropin_retries.log"), "a") as f:
                    f.write("==== %s\n%s\n" % (time.strftime("%Y-%m-%d %H:%M:%S"), text[-3000:]))
            except OSError:
                pass
            time.sleep(2.0)
            p = subprocess.run(*args, **kw)
    return p


def run_driver(tmp_path, name, *edits):
    nam, out = str(tmp_path / f"config_{name}.nam"), str(tmp_path / f"ecrad_meridian_{name}_out.nc")
    write_namelist(nam, *edits)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = _run([EXE, nam, MERIDIAN, out], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=900)
    return p, out


needs_exe = pytest.mark.skipif(not os.path.exists(EXE), reason="tests/_build/dropin/ecrad_hip has not been built (tools/build_dropin.py)")


@needs_exe
@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="a GPU is visible")
def test_without_a_device_the_host_prepares_its_tables_and_the_dropin_fails_loudly(tmp_path):
    """On a box without a GPU: the namelist is read, the reference's setup routines read and map every look-up table
    through the repo's netCDF module, and the first thing the drop-in does -- create its handle -- aborts the run through
    radiation_abort; there is no CPU path behind radiation()."""
    nam, out = str(tmp_path / "config_ecckd_tc.nam"), str(tmp_path / "ecrad_meridian_ecckd_tc_out.nc")
    write_namelist(nam)
    p = subprocess.run([EXE, nam, MERIDIAN, out], capture_output=True, text=True, cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="1"), timeout=900)
    text = p.stdout + p.stderr
    assert p.returncode != 0 and not os.path.exists(out)
    assert "OFFLINE ECRAD RADIATION SCHEME" in text
    assert "Reading NetCDF file" in text and "ecckd-1.0_lw_climate_fsck-32b_ckd-definition.nc" in text
    assert "Aerosol mapping:" in text and "Sea salt, bin 1" in text          # (string attributes of the aerosol file)
    assert "no usable MI355X device" in text


@needs_exe
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(TARGETS))
def test_reference_driver_with_the_dropin_reproduces_the_golden_file(tmp_path, name):
    family, edits = TARGETS[name]
    p, out = run_driver(tmp_path, name, family, edits)
    assert p.returncode == 0 and os.path.exists(out), (p.stdout + p.stderr)[-3000:]
    worst = {}
    with NcFile(os.path.join(GOLDEN_DIR, f"ecrad_meridian_{name}_out_REFERENCE.nc")) as g, NcFile(out) as o:
        names = list(g._f.variables)
        assert sorted(names) == sorted(o._f.variables), (sorted(names), sorted(o._f.variables))
        for v in names:
            ref, got = g.get(v), o.get(v)
            assert got.shape == ref.shape, (v, got.shape, ref.shape)
            assert g._f.variables[v].dimensions == o._f.variables[v].dimensions, v
            worst[v] = rel_err(got, ref)
    bad = {k: e for k, e in worst.items() if not e < FLOAT32_TOL}
    assert not bad, f"{name}: beyond float32 rounding: {bad}"
    print(name, "drop-in driver vs golden file: max", max(worst.values()))


OMP_EXE = os.path.join(ROOT, "tests", "_build", "dropin_omp", "ecrad_hip")


def _pool_report(text):
    """The line ECRAD_HIP_POOL_REPORT=1 makes the library print when the process ends (ecrad_amd/csrc/pool.hip: report_pools)."""
    m = re.search(r"ecrad_hip pool: devices (\d+) contexts (\d+) calls (\d+) max_in_flight (\d+) batches (\d+) calls_on_device(.*)", text)
    assert m, text[-2000:]
    return {"devices": int(m.group(1)), "contexts": int(m.group(2)), "calls": int(m.group(3)), "max_in_flight": int(m.group(4)),
            "batches": int(m.group(5)), "calls_on_device": dict((int(a), int(b)) for a, b in re.findall(r"(\d+):(\d+)", m.group(6)))}


@pytest.mark.skipif(not os.path.exists(OMP_EXE), reason="tests/_build/dropin_omp/ecrad_hip has not been built (tools/build_dropin.py --openmp)")
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["default", "tripleclouds", "ecckd_mcica"])
def test_openmp_driver_threads_call_the_dropin_concurrently(tmp_path, name):
    """The reference's radiation() is re-entrant and its driver calls it from `!$OMP PARALLEL DO` over blocks of columns
    (driver/ecrad_driver.F90:348; SURVEY.md 8(b) "Threading").  Here that driver is compiled WITH OpenMP around the drop-in
    (tools/build_dropin.py --openmp) and run on 16 threads with blocks of 2 columns: 16 calls of radiation() at once on the one
    handle, which the library runs together: the blocks that wait while a batch is on the device form the next batch (at least
    8 calls in flight at some point and fewer batches than calls, by the library's own count) -- and the output file still
    equals the reference's golden file to float32 rounding."""
    family, edits = TARGETS[name]
    nam, out = str(tmp_path / f"config_{name}.nam"), str(tmp_path / f"ecrad_meridian_{name}_out.nc")
    write_namelist(nam, family, edits)
    text = open(nam).read()
    assert len(re.findall(r"nblocksize\s*=\s*\d+", text)) == 1
    open(nam, "w").write(re.sub(r"nblocksize\s*=\s*\d+", "nblocksize = 2", text))
    env = dict(os.environ, OMP_NUM_THREADS="16", OMP_STACKSIZE="1G", ECRAD_HIP_CONTEXTS="16", ECRAD_HIP_DEVICES="1", ECRAD_HIP_POOL_REPORT="1")
    p = _run(f"ulimit -s unlimited; exec {OMP_EXE} {nam} {MERIDIAN} {out}", shell=True, capture_output=True, text=True,
                       cwd=str(tmp_path), env=env, timeout=900)
    assert p.returncode == 0 and os.path.exists(out), (p.stdout + p.stderr)[-3000:]
    pool = _pool_report(p.stdout + p.stderr)
    assert pool["contexts"] == 16 and pool["calls"] == 16 and pool["max_in_flight"] >= 8 and pool["batches"] < 16, pool
    worst = {}
    with NcFile(os.path.join(GOLDEN_DIR, f"ecrad_meridian_{name}_out_REFERENCE.nc")) as g, NcFile(out) as o:
        for v in g._f.variables:
            ref, got = g.get(v), o.get(v)
            assert got.shape == ref.shape, (v, got.shape, ref.shape)
            worst[v] = rel_err(got, ref)
    bad = {k: e for k, e in worst.items() if not e < FLOAT32_TOL}
    assert not bad, f"{name}: beyond float32 rounding: {bad}"
    print(name, f"OpenMP driver (16 threads, blocks of 2 columns, {pool['max_in_flight']} calls in flight at once) + drop-in vs golden file: max", max(worst.values()))


@pytest.mark.skipif(not os.path.exists(OMP_EXE), reason="tests/_build/dropin_omp/ecrad_hip has not been built (tools/build_dropin.py --openmp)")
@pytest.mark.gpu
def test_openmp_driver_with_the_reference_block_size_of_80_columns(tmp_path):
    """The reference's own test namelist sets nblocksize = 80 (test/ifs/configCY49R1_ecckd.nam:12).  5 120 synthetic columns
    (ecCKD-32 Tripleclouds with clouds and aerosols, written as a driver input file) through the UNCHANGED OpenMP driver +
    drop-in: 64 blocks of 80 columns on 16 threads against one block of 5 120 columns on one thread -- every variable of
    the output file identical, at least 8 calls in flight, and the driver's own timer (driver/ecrad_driver.F90:387-388) for both."""
    from bench import build_config
    from ecrad_amd.driver import save_inputs
    from ecrad_amd.synthetic import make_columns
    config, clear_sky, _ = build_config("tripleclouds_ecckd32")
    inputs = make_columns(config, 5120, clear_sky)
    inp = str(tmp_path / "inputs.nc")
    save_inputs(inp, config, *inputs[2:])
    write_namelist(str(tmp_path / "base.nam"), {"do_save_spectral_flux": "false", "iverbose": "1", "iverbosesetup": "0"})
    base = open(str(tmp_path / "base.nam")).read()
    outs, times = {}, {}
    for tag, nthreads, nblock in (("blocks", 16, 80), ("whole", 1, 5120)):
        nam, out = str(tmp_path / f"config_{tag}.nam"), str(tmp_path / f"out_{tag}.nc")
        # (nrepeat = 20: the driver repeats its block loop inside its timer, driver/ecrad_driver.F90:340-388, so that what each
        #  context allocates on its first call -- work arrays, page-locked staging -- is a twentieth of what is timed)
        open(nam, "w").write(re.sub(r"nrepeat\s*=\s*\d+", "nrepeat = 20", re.sub(r"nblocksize\s*=\s*\d+", f"nblocksize = {nblock}", base)))
        env = dict(os.environ, OMP_NUM_THREADS=str(nthreads), OMP_STACKSIZE="1G", ECRAD_HIP_CONTEXTS="16", ECRAD_HIP_DEVICES="1", ECRAD_HIP_POOL_REPORT="1")
        p = _run(f"ulimit -s unlimited; exec {OMP_EXE} {nam} {inp} {out}", shell=True, capture_output=True, text=True,
                 cwd=str(tmp_path), env=env, timeout=900)
        text = p.stdout + p.stderr
        assert p.returncode == 0 and os.path.exists(out), text[-3000:]
        pool = _pool_report(text)
        m = re.search(r"Time elapsed in radiative transfer:\s*([0-9.Ee+-]+)\s*seconds", text)
        assert m, text[-2000:]
        times[tag] = float(m.group(1)) / 20.0
        outs[tag] = out
        if tag == "blocks":
            assert pool["calls"] == 64 * 20 and pool["max_in_flight"] >= 8, pool
        else:
            assert pool["calls"] == 20 and pool["max_in_flight"] == 1, pool
    with NcFile(outs["blocks"]) as a, NcFile(outs["whole"]) as b:      # (the files differ in their time stamp only)
        names = list(a._f.variables)
        assert sorted(names) == sorted(b._f.variables) and len(names) >= 10
        for v in names:
            assert np.array_equal(a.get(v), b.get(v), equal_nan=True), v
    print("OpenMP driver + drop-in, 5120 columns, per repeat of 20: 64 blocks of 80 on 16 threads %.4f s (%.0f columns/s); one block on one thread %.4f s (%.0f columns/s)"
          % (times["blocks"], 5120 / times["blocks"], times["whole"], 5120 / times["whole"]))


# Targets of test/ifs/Makefile WITHOUT a golden file: (namelist family, change_namelist.sh arguments, the same as a Python-host config)
OTHER_TARGETS = {
    "test_lwscat": (RRTMG, {"do_lw_cloud_scattering": "true"}, ("rrtmg", "McICA", dict(do_lw_cloud_scattering=True))),
    "test_vec": (RRTMG, {"use_vectorizable_generator": "true"}, ("rrtmg", "McICA", dict(use_vectorizable_generator=True))),
    "test_spartacus": (RRTMG, {"sw_solver_name": '"SPARTACUS"', "lw_solver_name": '"SPARTACUS"', "do_3d_effects": "true",
                               "do_sw_delta_scaling_with_gases": "false"},
                       ("rrtmg", "SPARTACUS", dict(do_3d_effects=True, do_sw_delta_scaling_with_gases=False))),
    "test_spartacus_maxentr": (RRTMG, {"sw_solver_name": '"SPARTACUS"', "lw_solver_name": '"SPARTACUS"', "do_3d_effects": "true",
                                       "sw_entrapment_name": '"Maximum"', "do_sw_delta_scaling_with_gases": "false"},
                               ("rrtmg", "SPARTACUS", dict(do_3d_effects=True, i_3d_sw_entrapment=4, do_sw_delta_scaling_with_gases=False))),
    "test_ecckd_tc": ({}, {}, ("ecckd", "Tripleclouds", {})),
    "test_ecckd_noaer": ({}, {"use_aerosols": "false"}, ("ecckd", "Tripleclouds", dict(use_aerosols=False))),
    "test_ecckd_spartacus": ({}, {"sw_solver_name": '"SPARTACUS"', "lw_solver_name": '"SPARTACUS"', "do_3d_effects": "true"},
                             ("ecckd", "SPARTACUS", dict(do_3d_effects=True))),
}


@needs_exe
@pytest.mark.gpu
@pytest.mark.parametrize("target", sorted(OTHER_TARGETS))
def test_reference_driver_targets_without_golden_match_the_python_host(tmp_path, target):
    """The other targets of the reference's test suite: the reference's driver + drop-in (Fortran table preparation by the
    reference's own set-up routines, file I/O through the repo's netCDF module) against the Python host
    (ecrad_amd/spectral.py, tables.py: a restatement of that table preparation) -- two independent hosts, ONE device library,
    the same namelist.  Every variable of the output file, double precision, 1e-9: what differs between the two is only what
    the hosts hand to ecrad_hip_setup."""
    from ecrad_amd.driver import flux_to_output_dict
    from helpers import make_config, make_config_rrtmg, run_case
    family, edits, (fam, solver, kw) = OTHER_TARGETS[target]
    nam, out = str(tmp_path / f"config_{target}.nam"), str(tmp_path / "out.nc")
    write_namelist(nam, family, edits)
    text = open(nam).read()
    assert text.count("do_write_double_precision = false") == 1
    open(nam, "w").write(text.replace("do_write_double_precision = false", "do_write_double_precision = true"))
    p = _run([EXE, nam, MERIDIAN, out], capture_output=True, text=True, cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="1"), timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    kw = dict(do_save_spectral_flux=True, do_lw_aerosol_scattering=False, **kw)
    cfg = make_config_rrtmg(solver, **kw) if fam == "rrtmg" else make_config(solver, **kw)
    flux, th, rad = run_case(cfg, "hip")
    rad.close()
    want = flux_to_output_dict(cfg, th, flux)
    worst = {}
    with NcFile(out) as o:
        names = list(o._f.variables)
        assert len(names) >= 20
        for v in names:
            assert v in want, f"{target}: the Python host does not produce {v}"
            got, ref = o.get(v), np.asarray(want[v])
            assert got.shape == ref.shape, (v, got.shape, ref.shape)
            worst[v] = rel_err(got, ref)
    # (the longwave 3-D terms of SPARTACUS -- unpivoted 6x6 solves with a nearly singular matrix, radiation_spartacus_lw.F90:700-740 --
    #  amplify the last bits by which the two hosts' tables differ)
    tol = 1.0e-7 if solver == "SPARTACUS" else 1.0e-9
    bad = {k: e for k, e in worst.items() if not e < tol}
    assert not bad, f"{target}: Fortran host vs Python host: {bad}"
    print(target, "reference driver + drop-in vs Python host: max", max(worst.values()))


@needs_exe
@pytest.mark.gpu
def test_reference_driver_spartacus_target_matches_the_python_host(tmp_path, oracle_lib):
    """test_spartacus of test/ifs/Makefile (no golden file exists): the reference's driver + drop-in against the oracle run
    through the Python host on the same namelist -- two independent hosts (Fortran table preparation by the reference's
    own routines, Python restatement of it), one device library."""
    if not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    from ecrad_amd.driver import flux_to_output_dict
    from helpers import make_config_rrtmg, run_case
    edits = {"sw_solver_name": '"SPARTACUS"', "lw_solver_name": '"SPARTACUS"', "do_3d_effects": "true",
             "do_sw_delta_scaling_with_gases": "false"}
    nam, out = str(tmp_path / "config_spartacus.nam"), str(tmp_path / "out.nc")
    write_namelist(nam, RRTMG, edits)
    # (do_write_double_precision belongs to the driver group)
    text = open(nam).read()
    assert text.count("do_write_double_precision = false") == 1
    open(nam, "w").write(text.replace("do_write_double_precision = false", "do_write_double_precision = true"))
    p = _run([EXE, nam, MERIDIAN, out], capture_output=True, text=True, cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="1"), timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    cfg = make_config_rrtmg("SPARTACUS", do_3d_effects=True, do_sw_delta_scaling_with_gases=False, do_save_spectral_flux=True)
    flux, th, _ = run_case(cfg, oracle_lib.make_rrtmg_backend(cfg))
    want = flux_to_output_dict(cfg, th, flux)
    with NcFile(out) as o:
        for v in ("flux_up_lw", "flux_dn_lw", "flux_up_sw", "flux_dn_sw", "flux_dn_direct_sw", "flux_up_lw_clear", "flux_up_sw_clear",
                  "cloud_cover_sw", "cloud_cover_lw", "lw_derivative"):
            assert rel_err(o.get(v), np.asarray(want[v])) < 1.0e-6, v


IFS_EXE = os.path.join(ROOT, "tests", "_build", "dropin", "ecrad_ifs_hip")
IFS_BLOCKED_EXE = os.path.join(ROOT, "tests", "_build", "dropin", "ecrad_ifs_blocked_hip")
IFS_REF = os.path.join(ROOT, "tests", "_build", "reference", "ecrad_ifs_ref")


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(IFS_EXE) and os.path.exists(IFS_REF)), reason="the IFS-driver builds of tools/build_dropin.py are missing")
@pytest.mark.parametrize("solver", ["Tripleclouds", "McICA"])
def test_the_second_caller_ifs_radiation_scheme_through_the_dropin(tmp_path, solver):
    """radiation() has a second call site, ifs/radiation_scheme.F90:540: the reference's IFS-style driver
    (driver/ecrad_ifs_driver.F90 and its NPROMA-blocked twin: radiation_setup, effective radii and overlap decorrelation
    length from the IFS parametrisations, radiation_scheme) built with the drop-in module runs on the GPU and is compared with
    the same program built from the UNMODIFIED reference and run on the CPU of this box (the target `test_ifsdriver` of
    test/ifs/Makefile: net fluxes, double precision).  A cross-check in the sense of tests/test_oracle_vs_reference_build.py."""
    nam = str(tmp_path / "config_net.nam")
    write_namelist(nam, RRTMG, {"sw_solver_name": f'"{solver}"', "lw_solver_name": f'"{solver}"'})
    text = open(nam).read()
    assert text.count("do_write_double_precision = false") == 1 and text.count("do_save_net_fluxes = false") == 1
    open(nam, "w").write(text.replace("do_write_double_precision = false", "do_write_double_precision = true")
                             .replace("do_save_net_fluxes = false", "do_save_net_fluxes = true"))
    # The reference's IFS drivers allocate `land_frac` and never assign it (driver/ecrad_ifs_driver.F90:345,434;
    # ecrad_ifs_driver_blocked.F90:346,369): the land-sea mask that picks the droplet number concentration of the liquid
    # effective radius (ifs/liquid_effective_radius.F90:124) is whatever the heap holds.  A fresh CPU process gets zero pages
    # (sea everywhere); a process that has loaded the HIP runtime does not always, and then the ALL-SKY fluxes of a column or
    # two change by a few per cent -- in the unmodified reference too (it changes its own output under MALLOC_PERTURB_=165).
    # So: clear-sky variables must agree on every attempt; the all-sky ones on one of up to six attempts.
    def attempt(k):
        outs = {}
        for label, exe in (("hip", IFS_EXE), ("hip_blocked", IFS_BLOCKED_EXE), ("ref", IFS_REF)):
            out = str(tmp_path / f"{label}_{k}.nc")
            env = dict(os.environ, OMP_NUM_THREADS="1" if label != "ref" else "8", OMP_STACKSIZE="1G", MALLOC_PERTURB_="85")
            p = _run(f"ulimit -s unlimited; exec {exe} {nam} {MERIDIAN} {out}", shell=True, capture_output=True, text=True,
                               cwd=str(tmp_path), env=env, timeout=900)
            assert p.returncode == 0, label + ": " + (p.stdout + p.stderr)[-3000:]
            outs[label] = out
        bad = {}
        with NcFile(outs["hip"]) as h, NcFile(outs["hip_blocked"]) as hb, NcFile(outs["ref"]) as r:
            names = list(r._f.variables)
            assert len(names) >= 12 and sorted(names) == sorted(h._f.variables)
            for v in names:
                e1, e2 = rel_err(h.get(v), r.get(v)), rel_err(hb.get(v), h.get(v))       # GPU vs CPU reference; NPROMA blocking changes nothing
                if "clear" in v or v in ("pressure_hl", "flux_dn_sw_toa"):
                    assert e1 < 1.0e-6 and e2 < 1.0e-12, (v, e1, e2)
                elif not (e1 < 1.0e-6 and e2 < 1.0e-12):
                    bad[v] = (e1, e2)
        return bad
    for k in range(6):
        bad = attempt(k)
        if not bad:
            break
    assert not bad, bad


REF_EXE = os.path.join(ROOT, "tests", "_build", "reference", "ecrad_ref")
ALL_TARGETS = {**{k: (fam, ed) for k, (fam, ed) in TARGETS.items()}, **{k: (fam, ed) for k, (fam, ed, _) in OTHER_TARGETS.items()}}


def _run_both(tmp_path, nam, inp):
    outs = {}
    for label, exe, threads in (("hip", EXE, "1"), ("ref", REF_EXE, "8")):
        out = str(tmp_path / f"{label}_out.nc")
        env = dict(os.environ, OMP_NUM_THREADS=threads, OMP_STACKSIZE="1G")
        p = _run(f"ulimit -s unlimited; exec {exe} {nam} {inp} {out}", shell=True, capture_output=True, text=True,
                           cwd=str(tmp_path), env=env, timeout=1800)
        assert p.returncode == 0, label + ": " + (p.stdout + p.stderr)[-3000:]
        outs[label] = out
    return outs


def _double_precision_output(nam):
    text = open(nam).read()
    assert text.count("do_write_double_precision = false") == 1
    open(nam, "w").write(text.replace("do_write_double_precision = false", "do_write_double_precision = true"))


def _compare_files(outs, tol):
    worst = {}
    with NcFile(outs["hip"]) as h, NcFile(outs["ref"]) as r:
        names = list(r._f.variables)
        assert len(names) >= 20 and sorted(names) == sorted(h._f.variables)
        for v in names:
            a, b = h.get(v), r.get(v)
            assert a.shape == b.shape, v
            worst[v] = rel_err(a, b)
    bad = {k: e for k, e in worst.items() if not e < tol}
    assert not bad, bad
    return max(worst.values())


both_exes = pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(REF_EXE)), reason="tools/build_dropin.py [--reference] builds are missing")


@both_exes
@pytest.mark.gpu
@pytest.mark.parametrize("target", sorted(ALL_TARGETS))
def test_gpu_dropin_against_the_reference_executable_on_identical_netcdf_inputs(tmp_path, target):
    """north_star: "fluxes must match the reference CPU OpenMP path on identical netCDF inputs to <= 1e-6 relative (double
    precision)".  Literally that: the same namelist and the same input file through the reference's offline executable on this
    box's CPU (tests/_build/reference/ecrad_ref: ecRad 1.7.1 compiled unmodified, OpenMP) and through the same executable with the
    drop-in module (tests/_build/dropin/ecrad_hip, the MI355X), every variable of the two double-precision output files, for
    every target of test/ifs/Makefile.  (Both rest on the repo's netCDF library for file I/O: a cross-check by the task's rules,
    cf. tests/test_oracle_vs_reference_build.py.)"""
    family, edits = ALL_TARGETS[target]
    nam = str(tmp_path / f"config_{target}.nam")
    write_namelist(nam, family, edits)
    _double_precision_output(nam)
    worst = _compare_files(_run_both(tmp_path, nam, MERIDIAN), 1.0e-6)
    print(target, "GPU drop-in vs reference executable: max", worst)


@both_exes
@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["clear_homogeneous_ecckd32", "tripleclouds_ecckd32", "mcica_ecckd32", "mcica_rrtmg"])
def test_gpu_dropin_against_the_reference_executable_on_synthetic_ifs_shaped_columns(tmp_path, workload):
    """The same on 4 096 of the synthetic IFS-shaped columns bench.py times (ecrad_amd/synthetic.py), written to a
    netCDF file the way the reference's own save_inputs would (ecrad_amd.driver.save_inputs): identical file, two executables."""
    from bench import build_config
    from ecrad_amd.driver import save_inputs
    from ecrad_amd.interface import setup_radiation
    from ecrad_amd.synthetic import make_columns
    config, clear_sky, desc = build_config(workload)
    setup_radiation(config)
    inputs = make_columns(config, 4096, clear_sky)
    inp = str(tmp_path / "synthetic.nc")
    save_inputs(inp, config, *inputs[2:])
    solver = f'"{desc["sw_solver"]}"'
    edits = {"sw_solver_name": solver, "lw_solver_name": solver, "do_save_spectral_flux": "false",
             "use_aerosols": "true" if config.use_aerosols else "false"}
    nam = str(tmp_path / "config.nam")
    write_namelist(nam, RRTMG if desc["rrtmg"] else {}, edits)
    _double_precision_output(nam)
    worst = _compare_files(_run_both(tmp_path, nam, inp), 1.0e-6)
    print(workload, "4096 synthetic columns, GPU drop-in vs reference executable: max", worst)


@needs_exe
@pytest.mark.gpu
def test_the_dropins_own_timer_at_100000_columns_in_one_call(tmp_path):
    """The boundary as the reference's offline driver uses it at the size of BASELINE configs[1]: 100 000 synthetic clear-sky
    columns in a netCDF file, ONE block (nblocksize = 100 000), i.e. one radiation() call on host arrays per repeat -- the
    drop-in pipelines it over PCIe as column tiles (pipeline.hip: radiation_host_pipelined) -- timed by the driver's own timer
    (driver/ecrad_driver.F90:387-388: "Time elapsed in radiative transfer").  Printed; the bound asserted is a loose one
    (half of what one MI355X box gave: 2.1 M columns/s, bench.py end_to_end_host), the number is for the log."""
    from bench import build_config
    from ecrad_amd.driver import save_inputs
    from ecrad_amd.interface import setup_radiation
    from ecrad_amd.synthetic import make_columns
    ncol = 100000
    config, clear_sky, desc = build_config("clear_homogeneous_ecckd32")
    setup_radiation(config)
    inputs = make_columns(config, ncol, clear_sky)
    inp = str(tmp_path / "synthetic100k.nc")
    save_inputs(inp, config, *inputs[2:])
    del inputs
    nam, out = str(tmp_path / "config.nam"), str(tmp_path / "out.nc")
    env = dict(os.environ, OMP_NUM_THREADS="1", OMP_STACKSIZE="1G")

    def timed(nrepeat):
        write_namelist(nam, {"sw_solver_name": '"Homogeneous"', "lw_solver_name": '"Homogeneous"', "use_aerosols": "false",
                             "do_save_spectral_flux": "false"})
        text = open(nam).read()
        text = re.sub(r"nblocksize\s*=\s*\d+", f"nblocksize = {ncol}", text)
        text = re.sub(r"nrepeat\s*=\s*\d+", f"nrepeat = {nrepeat}", text)
        assert f"nblocksize = {ncol}" in text and f"nrepeat = {nrepeat}" in text
        open(nam, "w").write(text)
        p = _run(f"ulimit -s unlimited; exec {EXE} {nam} {inp} {out}", shell=True, capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=1800)
        assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
        m = re.search(r"Time elapsed in radiative transfer:\s*([0-9.eE+-]+)", p.stdout + p.stderr)
        assert m, (p.stdout + p.stderr)[-2000:]
        return float(m.group(1))

    # (the first call of a process allocates the device arrays and touches the freshly allocated flux arrays of the driver for
    #  the first time: it is timed on its own, the calls after it by the difference between a run of one and a run of five)
    first, five = timed(1), timed(5)
    seconds = (five - first) / 4
    print(f"reference driver + drop-in, {ncol} clear-sky columns in one call, the driver's own timer: first call {first * 1e3:.1f} ms, "
          f"then {seconds * 1e3:.1f} ms per call -> {ncol / seconds:.0f} columns/s (host arrays, PCIe-inclusive)")
    assert ncol / seconds > 1.0e6
    with NcFile(out) as f:
        up = f.get("flux_up_lw")
        assert up.shape[0] == ncol and np.isfinite(up).all() and up.min() > 50.0


# ---- SPARTACUS with two regions (config%nregions = 2, radiation_config.F90:268) ---------------------------------------------------
_SP2 = {"sw_solver_name": '"SPARTACUS"', "lw_solver_name": '"SPARTACUS"', "n_regions": "2"}
TWO_REGION_CASES = {
    # name: (namelist family, edits, spectra compared)
    "ecckd_1d": ({}, dict(_SP2, do_3d_effects="false"), ("sw", "lw")),
    "rrtmg_1d": (RRTMG, dict(_SP2, do_3d_effects="false", do_sw_delta_scaling_with_gases="false"), ("sw", "lw")),
    "ecckd_1d_beta_overlap": ({}, dict(_SP2, do_3d_effects="false", use_beta_overlap="true"), ("sw", "lw")),
    "ecckd_3d_explicit_entrapment": ({}, dict(_SP2, do_3d_effects="true"), ("sw",)),
    "rrtmg_3d_maximum_entrapment": (RRTMG, dict(_SP2, do_3d_effects="true", sw_entrapment_name='"Maximum"',
                                                do_sw_delta_scaling_with_gases="false"), ("sw",)),
    "ecckd_3d_zero_entrapment": ({}, dict(_SP2, do_3d_effects="true", sw_entrapment_name='"Zero"'), ("sw",)),
}


@both_exes
@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(TWO_REGION_CASES))
def test_gpu_dropin_against_the_reference_executable_with_two_spartacus_regions(tmp_path, case):
    """config%nregions = 2 (one homogeneous cloudy region: radiation_regions.F90:105-110, radiation_overlap.F90:169-175,
    radiation_spartacus_sw.F90:499, :1184): the reference's offline executable on this box's CPU against the same executable
    with the drop-in, every variable of the two double-precision output files.  The HIP kernels run two regions through their
    three-region arrays with an empty third region (kernel_prep.hip: tripleclouds_prep_kernel).

    With 3-D effects the LONGWAVE of the reference itself is not usable at nregions = 2: radiation_spartacus_lw.F90:509-520 tests
    edge_length(3,jlev), which only the nregions > 2 branch (:457) assigns, and then stores into transfer_rate(1,3) / (3,1) of
    a 2 x 2 array -- the unmodified reference returns longwave fluxes of 1e27 W m-2 on the test/ifs profiles.  Those cases
    compare the shortwave variables (the same statements exist at radiation_spartacus_sw.F90:583-597; there the reference's
    output is finite and is reproduced) and check that the drop-in's longwave fluxes are physical."""
    family, edits, spectra = TWO_REGION_CASES[case]
    nam = str(tmp_path / f"config_{case}.nam")
    write_namelist(nam, family, edits)
    _double_precision_output(nam)
    outs = _run_both(tmp_path, nam, MERIDIAN)
    worst = {}
    with NcFile(outs["hip"]) as h, NcFile(outs["ref"]) as r:
        names = list(r._f.variables)
        assert len(names) >= 20 and sorted(names) == sorted(h._f.variables)
        for v in names:
            a, b = h.get(v), r.get(v)
            assert a.shape == b.shape, v
            if "lw" in v and "lw" not in spectra:
                assert np.all(np.isfinite(a)) and (v == "lw_derivatives" or (a.min() >= 0.0 and a.max() < 1000.0)), v
                continue
            worst[v] = rel_err(a, b)
    bad = {k: e for k, e in worst.items() if not e < 1.0e-6}
    assert not bad, bad
    print(case, "two regions, GPU drop-in vs reference executable: max", max(worst.values()))


# ---- the reference's two other test directories through the two executables ---------------------------------------------
def _suite_namelist(src, dst, radiation_edits=None, driver_edits=None):
    text = open(src).read()
    drv, rad = text.split("&radiation\n", 1)
    def apply(block, edits):
        for k, v in (edits or {}).items():
            pat = re.compile(r"^(\s*)" + re.escape(k) + r"\s*=[^,\n]*,?", re.M)
            block = pat.sub(lambda m: f"{m.group(1)}{k} = {v},", block, count=1) if pat.search(block) else f"{k} = {v},\n" + block
        return block
    rad = apply(rad, dict(radiation_edits or {}, directory_name=f'"{DATA_DIR}"'))
    head, body = drv.split("&radiation_driver\n", 1)
    body = apply(body, dict(driver_edits or {}, do_write_double_precision="true"))
    open(dst, "w").write(head + "&radiation_driver\n" + body + "&radiation\n" + rad)


SUITES = {
    # test/ckdmip: 50 clear-sky CKDMIP profiles (54 levels, gases as "<gas>_mole_fraction_fl"), ecCKD and RRTMG
    "ckdmip_ecckd": ("ckdmip/ckdmip.nam", "ckdmip/ckdmip_evaluation1_concentrations_present_reduced.nc", {}, {}),
    "ckdmip_rrtmg": ("ckdmip/ckdmip.nam", "ckdmip/ckdmip_evaluation1_concentrations_present_reduced.nc", {"gas_model_name": '"RRTMG-IFS"'},
                     {"cos_solar_zenith_angle": "0.1"}),
    # test/i3rc: the I3RC cumulus profile (164 levels) over 46 solar zenith angles, SPARTACUS with 3-D effects on RRTMG's spectra
    "i3rc_maximum": ("i3rc/i3rc.nam", "i3rc/i3rc_mls_cumulus.nc", {}, {}),
    "i3rc_explicit": ("i3rc/i3rc.nam", "i3rc/i3rc_mls_cumulus.nc", {"sw_entrapment_name": '"Explicit"'}, {}),
    "i3rc_1d": ("i3rc/i3rc.nam", "i3rc/i3rc_mls_cumulus.nc", {"do_3d_effects": "false"}, {}),
    "i3rc_tripleclouds": ("i3rc/i3rc.nam", "i3rc/i3rc_mls_cumulus.nc", {"sw_solver_name": '"Tripleclouds"', "lw_solver_name": '"Tripleclouds"'}, {}),
    "i3rc_mcica": ("i3rc/i3rc.nam", "i3rc/i3rc_mls_cumulus.nc", {"sw_solver_name": '"McICA"', "lw_solver_name": '"McICA"'}, {}),
}


@both_exes
@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(SUITES))
def test_gpu_dropin_against_the_reference_executable_on_the_other_reference_suites(tmp_path, case):
    """test/ckdmip and test/i3rc of the reference (input files byte-identical under tests/golden/; the namelists are the ones
    tests/test_reference_suites.py uses): other level counts (54, 164), gases as scalars / under other variable names, black
    surface and solar zenith angle from the driver namelist, cloud effective sizes from the file -- read by the reference's own
    input reader in both programs."""
    nam_src, inp, rad_edits, drv_edits = SUITES[case]
    nam = str(tmp_path / "config.nam")
    _suite_namelist(os.path.join(GOLDEN_DIR, nam_src), nam, rad_edits, drv_edits)
    outs = _run_both(tmp_path, nam, os.path.join(GOLDEN_DIR, inp))
    worst = {}
    with NcFile(outs["hip"]) as h, NcFile(outs["ref"]) as r:
        names = list(r._f.variables)
        assert len(names) >= 5 and sorted(names) == sorted(h._f.variables)
        for v in names:
            worst[v] = rel_err(h.get(v), r.get(v))
    bad = {k: e for k, e in worst.items() if not e < 1.0e-6}
    assert not bad, bad
    print(case, "GPU drop-in vs reference executable: max", max(worst.values()))


# ---- a single-precision host (the reference built with -DPARKIND1_SINGLE, jprb = real32: how the IFS runs operationally) ---
SP_EXE = os.path.join(ROOT, "tests", "_build", "dropin_sp", "ecrad_hip")
SP_REF = os.path.join(ROOT, "tests", "_build", "reference_sp", "ecrad_ref")


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(SP_EXE) and os.path.exists(SP_REF) and os.path.exists(EXE)),
                    reason="tools/build_dropin.py --single [--reference] builds are missing")
@pytest.mark.parametrize("target", ["ecckd_mcica", "test_ecckd_tc", "default", "tripleclouds", "test_spartacus"])
def test_single_precision_host_through_the_dropin(tmp_path, target):
    """In a single-precision build of the host the tables cross the boundary as double copies made once at set-up
    (radiation_hip_interface.F90: dloc / finish_copies; the RRTMG module tables likewise), the real32 arrays of a call go to
    ecrad_hip_radiation_f32 as they are (include/ecrad_hip.h: the library widens the columns of the call), and the device
    arithmetic stays double.  So the single-precision host + GPU must equal the DOUBLE-precision host + GPU up to the rounding of its inputs and
    outputs to float (5e-5 here: McICA turns a rounded cloud fraction into a different sub-column now and then; SPARTACUS, whose solver then runs in float as the reference's own single-precision build
    does, 2e-3), and must be at least as close to the double-precision result as the reference's own single-precision CPU
    executable is."""
    family, edits = ALL_TARGETS[target]
    nam = str(tmp_path / "config.nam")
    write_namelist(nam, family, edits)
    _double_precision_output(nam)
    outs = {}
    for label, exe, threads in (("sp_hip", SP_EXE, "1"), ("dp_hip", EXE, "1"), ("sp_ref", SP_REF, "8")):
        out = str(tmp_path / f"{label}.nc")
        p = _run(f"ulimit -s unlimited; exec {exe} {nam} {MERIDIAN} {out}", shell=True, capture_output=True, text=True,
                           cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS=threads, OMP_STACKSIZE="1G"), timeout=900)
        assert p.returncode == 0, label + ": " + (p.stdout + p.stderr)[-3000:]
        outs[label] = out
    spartacus = "spartacus" in target
    broadband = ("flux_up_lw", "flux_dn_lw", "flux_up_sw", "flux_dn_sw", "flux_dn_direct_sw", "flux_up_lw_clear", "flux_up_sw_clear")
    with NcFile(outs["sp_hip"]) as a, NcFile(outs["dp_hip"]) as b, NcFile(outs["sp_ref"]) as c:
        assert sorted(a._f.variables) == sorted(b._f.variables)
        e_hip = {v: rel_err(a.get(v), b.get(v)) for v in broadband}
        e_ref = {v: rel_err(c.get(v), b.get(v)) for v in broadband}
    print(target, "single-precision host + GPU vs double:", max(e_hip.values()), "; the reference's single-precision CPU run vs double:", max(e_ref.values()))
    print(target, "per variable (HIP sp vs dp, reference sp vs dp):", {v: (float(f"{e_hip[v]:.3g}"), float(f"{e_ref[v]:.3g}")) for v in broadband})
    for v in broadband:
        if spartacus and v in ("flux_up_lw", "flux_dn_lw"):
            # All-sky longwave with 3-D effects: in single precision the reference's own formulation amplifies the last bit (its sp run
            # is itself percents away from its dp run in a few layers), so the HIP path is not held to 2e-3 here but to the reference's
            # own envelope: finite everywhere and no further from double than a few times what the reference's sp build is.
            with NcFile(outs["sp_hip"]) as a:
                assert np.all(np.isfinite(a.get(v))), v
            assert e_hip[v] <= max(LW_SP_ENVELOPE * e_ref[v], 5.0e-2), (v, e_hip[v], e_ref[v])
            continue
        assert e_hip[v] < (2.0e-3 if spartacus else 5.0e-5), (v, e_hip[v])
        assert e_hip[v] <= max(2.0 * e_ref[v], 1.0e-6), (v, e_hip[v], e_ref[v])


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(EXE), reason="tests/_build/dropin/ecrad_hip has not been built (tools/build_dropin.py)")
def test_dropin_driver_writes_netcdf4_when_asked(tmp_path):
    """`do_write_hdf5 = true` in the driver's namelist (driver/ecrad_driver.F90:400 -> easy_netcdf's is_hdf5_file): the drop-in
    executable's output is a netCDF-4 / HDF5 file written by the repo's netcdf module (nc_classic.c: ecnc_h5_enddef), which the
    HDF5 library of the image reads back with the numbers of the classic file of the same run."""
    from test_hdf5_output import H5
    h5 = H5()
    family, edits = TARGETS["tripleclouds"]
    outs = {}
    for tag, hdf5 in (("classic", "false"), ("nc4", "true")):
        nam, out = str(tmp_path / f"{tag}.nam"), str(tmp_path / f"{tag}_out.nc")
        write_namelist(nam, family, edits)
        text = open(nam).read()
        open(nam, "w").write(text.replace("&radiation_driver\n", f"&radiation_driver\ndo_write_hdf5 = {hdf5},\n", 1))
        p = _run(f"ulimit -s unlimited; exec {EXE} {nam} {MERIDIAN} {out}", shell=True, capture_output=True, text=True,
                 cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="1", OMP_STACKSIZE="1G"), timeout=900)
        assert p.returncode == 0 and os.path.exists(out), (p.stdout + p.stderr)[-3000:]
        outs[tag] = out
    assert open(outs["classic"], "rb").read(3) == b"CDF" and open(outs["nc4"], "rb").read(4) == b"\x89HDF"
    f = h5.open(outs["nc4"])
    with NcFile(outs["classic"]) as c:
        names = set(c._f.variables)
        assert len(names) >= 20 and names <= set(h5.names(f))
        for n in names:
            _, a, _ = h5.read(f, n)
            assert np.array_equal(a, np.asarray(c.get(n), dtype=np.float64)), n
    h5.h5.H5Fclose(f)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(EXE), reason="tests/_build/dropin/ecrad_hip has not been built (tools/build_dropin.py)")
def test_dropin_driver_reads_a_netcdf4_input_file(tmp_path):
    """The reference's unmodified driver + the drop-in, handed its input as a netCDF-4 / HDF5 file (test/ifs/ecrad_meridian.nc rewritten by
    ecrad_amd/hdf5file.py; easy_netcdf.F90:133-200 opens whatever the netCDF library opens): the repo's netcdf module reads it through the HDF5
    library (nc_classic.c: ecnc_h5_open), and every variable of the output equals the run on the classic file."""
    from scipy.io import netcdf_file
    from ecrad_amd.hdf5file import write_nc4
    with netcdf_file(MERIDIAN, "r", mmap=False) as nc:
        dims = {d: (n if n is not None else nc.variables["pressure_hl"].shape[0]) for d, n in nc.dimensions.items()}
        variables = {name: (v.dimensions, np.array(v.data).reshape(v.shape)) for name, v in nc.variables.items()}
    inp4 = str(tmp_path / "meridian4.nc")
    write_nc4(inp4, dims, variables, double=True)
    assert open(inp4, "rb").read(4) == b"\x89HDF"
    family, edits = TARGETS["tripleclouds"]
    outs = {}
    for tag, inp in (("classic", MERIDIAN), ("nc4", inp4)):
        nam, out = str(tmp_path / f"{tag}.nam"), str(tmp_path / f"{tag}_out.nc")
        write_namelist(nam, family, edits)
        p = _run(f"ulimit -s unlimited; exec {EXE} {nam} {inp} {out}", shell=True, capture_output=True, text=True,
                 cwd=str(tmp_path), env=dict(os.environ, OMP_NUM_THREADS="1", OMP_STACKSIZE="1G"), timeout=900)
        assert p.returncode == 0 and os.path.exists(out), (p.stdout + p.stderr)[-3000:]
        outs[tag] = out
    with NcFile(outs["classic"]) as a, NcFile(outs["nc4"]) as b:
        names = list(a._f.variables)
        assert len(names) >= 20 and sorted(names) == sorted(b._f.variables)
        for v in names:
            assert np.array_equal(a.get(v), b.get(v), equal_nan=True), v


SP_OMP_EXE = os.path.join(ROOT, "tests", "_build", "dropin_sp_omp", "ecrad_hip")


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(SP_OMP_EXE) and os.path.exists(SP_EXE)), reason="tools/build_dropin.py --single [--openmp] builds are missing")
def test_single_precision_openmp_driver_blocks_of_80_run_concurrently(tmp_path):
    """The -DPARKIND1_SINGLE drop-in under the driver compiled WITH OpenMP: 5 120 synthetic columns as 64 blocks of 80 (the
    reference's nblocksize, test/ifs/configCY49R1_ecckd.nam:12) on 16 threads against the serial single-precision drop-in over
    one block of 5 120 -- every variable identical, and at least 8 calls of ecrad_hip_radiation_f32 in flight at once by the
    library's own count (round 4's wrapper took the calls of a single-precision host one at a time, inside an OpenMP critical
    section, converting whole ncol-sized arrays per call)."""
    from bench import build_config
    from ecrad_amd.driver import save_inputs
    from ecrad_amd.synthetic import make_columns
    config, clear_sky, _ = build_config("tripleclouds_ecckd32")
    inputs = make_columns(config, 5120, clear_sky)
    inp = str(tmp_path / "inputs.nc")
    save_inputs(inp, config, *inputs[2:])
    write_namelist(str(tmp_path / "base.nam"), {"do_save_spectral_flux": "false", "iverbose": "1", "iverbosesetup": "0"})
    base = open(str(tmp_path / "base.nam")).read()
    outs, times = {}, {}
    for tag, exe, nthreads, nblock in (("blocks", SP_OMP_EXE, 16, 80), ("whole", SP_EXE, 1, 5120)):
        nam, out = str(tmp_path / f"config_{tag}.nam"), str(tmp_path / f"out_{tag}.nc")
        open(nam, "w").write(re.sub(r"nrepeat\s*=\s*\d+", "nrepeat = 10", re.sub(r"nblocksize\s*=\s*\d+", f"nblocksize = {nblock}", base)))
        env = dict(os.environ, OMP_NUM_THREADS=str(nthreads), OMP_STACKSIZE="1G", ECRAD_HIP_CONTEXTS="16", ECRAD_HIP_DEVICES="1", ECRAD_HIP_POOL_REPORT="1")
        p = _run(f"ulimit -s unlimited; exec {exe} {nam} {inp} {out}", shell=True, capture_output=True, text=True,
                 cwd=str(tmp_path), env=env, timeout=900)
        text = p.stdout + p.stderr
        assert p.returncode == 0 and os.path.exists(out), text[-3000:]
        pool = _pool_report(text)
        m = re.search(r"Time elapsed in radiative transfer:\s*([0-9.Ee+-]+)\s*seconds", text)
        assert m, text[-2000:]
        times[tag] = float(m.group(1)) / 10.0
        outs[tag] = out
        if tag == "blocks":
            assert pool["calls"] == 64 * 10 and pool["max_in_flight"] >= 8, pool
    with NcFile(outs["blocks"]) as a, NcFile(outs["whole"]) as b:
        names = list(a._f.variables)
        assert sorted(names) == sorted(b._f.variables) and len(names) >= 10
        for v in names:
            assert np.array_equal(a.get(v), b.get(v), equal_nan=True), v
    print("single-precision OpenMP driver + drop-in, 5120 columns, per repeat of 10: 64 blocks of 80 on 16 threads %.4f s (%.0f columns/s); one block %.4f s (%.0f columns/s)"
          % (times["blocks"], 5120 / times["blocks"], times["whole"], 5120 / times["whole"]))


@pytest.mark.gpu
def test_zz_gpu_processes_started_without_retries():
    """Last test of the file: how often _run had to start a GPU process twice (see STARTUP_SIGNATURES).  Zero is the rule;
    one or two are reported (stderr, gpurun_out/dropin_retries.log) as the box's; more than that is a defect."""
    if RETRIES:
        print("GPU processes that had to be started twice: %d\n%s" % (len(RETRIES), "\n----\n".join(RETRIES)))
    assert len(RETRIES) <= 2, RETRIES
