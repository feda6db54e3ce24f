"""(GPU box) Repeat the NPROMA-blocked IFS driver through the drop-in and compare with the unblocked one: which runs differ, where."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_fortran_dropin as t
from ecrad_amd.ncfile import NcFile
tmp = "/tmp/diag_blocked"; os.makedirs(tmp, exist_ok=True)
nam = os.path.join(tmp, "net.nam")
t.write_namelist(nam, t.RRTMG, {"sw_solver_name": '"Tripleclouds"', "lw_solver_name": '"Tripleclouds"'})
text = open(nam).read().replace("do_write_double_precision = false", "do_write_double_precision = true").replace("do_save_net_fluxes = false", "do_save_net_fluxes = true")
open(nam, "w").write(text)
def run(exe, out, env_extra):
    env = dict(os.environ, OMP_NUM_THREADS="1", **env_extra)
    p = subprocess.run(f"ulimit -s unlimited; exec {exe} {nam} {t.MERIDIAN} {out}", shell=True, capture_output=True, text=True, cwd=tmp, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    return p.stdout
out0 = run(t.IFS_EXE, os.path.join(tmp, "ref.nc"), {})
print([l for l in out0.split("\n") if "block" in l.lower() or "nproma" in l.lower()][:5])
with NcFile(os.path.join(tmp, "ref.nc")) as r:
    ref = {v: r.get(v) for v in r._f.variables}
for label, extra in (("default", {}), ("no_overlap", {"ECRAD_NO_SPECTRA_OVERLAP": "1"})):
    nbad = 0
    for i in range(15):
        run(t.IFS_BLOCKED_EXE, os.path.join(tmp, "b.nc"), extra)
        with NcFile(os.path.join(tmp, "b.nc")) as b:
            bad = {}
            for v in ref:
                d = np.abs(b.get(v) - ref[v]) / (np.abs(ref[v]).max() + 1e-300)
                if d.max() > 1e-12:
                    cols = np.unique(np.nonzero(d > 1e-12)[0])
                    bad[v] = (float(d.max()), cols[:12].tolist())
        if bad:
            nbad += 1
            print(label, "run", i, "differs:", bad)
    print(label, "runs differing:", nbad, "of 15")
