"""Golden vectors for RRTMG gas optics (SURVEY.md section 8a, row a6 -- not built yet): produced by the
reference's own ifsrrtm routines compiled from where they lie (oracle/build_ref_rrtm.sh,
oracle/ref_rrtm_wrappers.F90, oracle/make_rrtm_golden.py).  These tests pin the fixture itself -- shapes,
physical sanity, and, where the reference is available, that the committed file is what the reference's
code produces -- so that the implementation of row a6 starts from a trusted target."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "rrtmg_gas_optics.npz")


@pytest.fixture(scope="module")
def g():
    return np.load(FIXTURE)


def test_shapes_and_conventions(g):
    ncol = len(g["columns"])
    nlev = g["q"].shape[0]
    assert g["pressure_hl"].shape == (nlev + 1, ncol) and g["temperature_hl"].shape == (nlev + 1, ncol)
    assert g["od_lw"].shape == (ncol, nlev, 140) and g["pfrac"].shape == (nlev, 140, ncol)
    assert g["od_sw"].shape == (112, nlev, ncol) and g["ssa_sw"].shape == (112, nlev, ncol)
    assert g["incsol"].shape == (112, ncol)
    # inputs run from the top of the atmosphere down, outputs from the surface up (rrtm_prepare_gases.F90)
    assert np.all(np.diff(g["pressure_hl"], axis=0) > 0)


def test_physical_sanity(g):
    # Planck fractions sum to one within each of the 16 bands (yoerrtm.F90:57-76 g-points per band)
    ngb = [10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2]
    edges = np.cumsum([0] + ngb)
    pf = g["pfrac"]
    for b in range(16):
        s = pf[:, edges[b]:edges[b + 1], :].sum(axis=1)
        # (bands 12 and 15 have no absorber above the tropopause: rrtm_taumol12/15 return zero fractions there)
        ok = np.isclose(s, 1.0, atol=2e-4) | ((s == 0.0) if b + 1 in (12, 15) else False)
        assert np.all(ok), b
    # night-time columns get no incoming solar flux, the others the RRTMG solar constant
    night = g["cos_sza"] <= 0
    tot = g["incsol"].sum(axis=0)
    assert np.all(tot[night] == 0) and np.allclose(tot[~night], 1368.22, atol=1e-2)
    assert np.all(g["od_sw"][:, :, ~night] >= 0) and np.all((g["ssa_sw"] >= 0) & (g["ssa_sw"] <= 1))
    # a few longwave g-points are known to go slightly negative (radiation_ifs_rrtm.F90:492-496): the
    # caller clamps them; everything else is non-negative and finite
    assert np.isfinite(g["od_lw"]).all() and (g["od_lw"] < 0).mean() < 0.01


@pytest.mark.skipif(not os.path.isdir("/root/reference/ifsrrtm"), reason="needs the reference checkout")
def test_fixture_is_what_the_reference_produces(g, tmp_path):
    """Rebuild the reference routines and regenerate: bit-identical to the committed vectors."""
    lib = os.path.join(ROOT, "oracle", "_ref", "libecrad_refrrtm.so")
    if not os.path.exists(lib):
        subprocess.run(["bash", os.path.join(ROOT, "oracle", "build_ref_rrtm.sh")], check=True, capture_output=True)
    before = {k: g[k].copy() for k in g.files}
    backup = open(FIXTURE, "rb").read()
    try:
        subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_rrtm_golden.py")], check=True, capture_output=True)
        again = np.load(FIXTURE)
        for k, v in before.items():
            assert np.array_equal(v, again[k]), k
    finally:
        open(FIXTURE, "wb").write(backup)
