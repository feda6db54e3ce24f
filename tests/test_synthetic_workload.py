"""The synthetic bench workload as a tested object: `make_columns()` at the bench's 100 000 columns for the
workloads bench.py times (BASELINE.json configs[1]-[4] and the north-star shape), the HIP path (device-memory mode, as
timed) against the oracle on the first 2 048 columns.

Broadband profiles, derivatives and cloud cover must agree to 1e-8 (bar: 1e-6).  Per-g-point / per-band / canopy
surface and TOA values must agree to the bar itself, 1e-6, and where they differ by more than 1e-8 the difference
must be of the size the reference's OWN formulas produce when only the rounding of a*b+c changes: the oracle is run a
second time compiled with floating-point contraction (oracle/Makefile: fma), and the largest HIP-vs-oracle difference
of a field may not exceed 30x the largest fma-vs-plain difference of the oracle on the same columns.  (Found while
naming the 3e-8 that bench.py reported in round 1: `sw_dn_diffuse_surf_g`, g-points 3-6 of the 32-term shortwave
model; the stage arrays od/ssa agree to 2e-15 there and the difference arises in the
Meador-Weaver direct-beam terms (radiation_two_stream.F90:519-535), whose bracket cancels to O(od) for thin layers
and is divided by 1 - (k mu0)^2, which passes through zero when k mu0 = 1 inside a column.  With gfortran's default
-ffp-contract=fast the reference itself moves by the same amount.)"""
import ctypes as C

import numpy as np
import pytest

from bench import build_config, first_columns, oracle_backend
from ecrad_amd.interface import Radiation
from ecrad_amd.synthetic import make_columns
from ecrad_amd.types import Flux

pytestmark = pytest.mark.gpu
NCOL, NCHECK = 100000, 2048
TOL, TOL_SPECTRAL = 1.0e-8, 1.0e-6


@pytest.mark.parametrize("workload", ["clear_homogeneous_ecckd32", "tripleclouds_ecckd32", "mcica_rrtmg", "tripleclouds_ecckd64"])
def test_synthetic_bench_columns_match_oracle(workload, oracle_lib):
    import torch
    from ecrad_amd.device import DeviceCase
    config, clear_sky, desc = build_config(workload)
    if desc["rrtmg"] and not oracle_lib.have_ref_rrtm():
        pytest.skip("oracle/_ref/libecrad_refrrtm.so (the reference's RRTMG routines) has not been built")
    rad = Radiation(config, backend="hip")
    inputs = make_columns(config, NCOL, clear_sky)
    n, nlev, sl, th, gas, cloud, aer = inputs
    sample = first_columns(inputs, NCHECK)
    flux = Flux.allocate(config, n, nlev)
    case = DeviceCase(config, n, nlev, sl, th, gas, cloud, aer, flux)
    st = rad.lib.ecrad_hip_radiation(rad.handle, n, nlev, 1, n, C.byref(case.inputs), C.byref(case.flux))
    assert st == 0, rad.lib.ecrad_hip_last_error(rad.handle).decode()
    rad.lib.ecrad_hip_synchronize(rad.handle)
    torch.cuda.synchronize()
    case.flux_to_host(flux)
    del case

    config2, _, _ = build_config(workload)
    orad = Radiation(config2, backend=oracle_backend(config2)[0])
    oflux = Flux.allocate(config2, NCHECK, nlev)
    orad.radiation(NCHECK, nlev, 1, NCHECK, *sample[2:], oflux)

    # the formulas' own sensitivity: the oracle with contraction against the plain oracle, same columns
    frad = Radiation(config2, backend=(oracle_lib.make_rrtmg_backend(config2, inner=oracle_lib.make_fma_variant_backend())
                                       if desc["rrtmg"] else oracle_lib.make_fma_variant_backend()))
    fflux = Flux.allocate(config2, NCHECK, nlev)
    frad.radiation(NCHECK, nlev, 1, NCHECK, *sample[2:], fflux)

    report = []
    for name, ref in oflux.arrays.items():
        got = flux.arrays[name]
        got = got[..., :NCHECK] if got.shape[-1] == n else got[:NCHECK]
        assert np.all(np.isfinite(got)), name
        scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max() + 1e-300)
        err = np.abs(got - ref) / scale
        idx = np.unravel_index(int(np.argmax(err)), err.shape)
        sens = float((np.abs(fflux.arrays[name] - ref) / scale).max())
        report.append((float(err[idx]), name, tuple(int(i) for i in idx), sens))
        spectral = name.endswith(("_g", "_band", "_canopy"))
        assert err[idx] <= (TOL_SPECTRAL if spectral else TOL), (name, idx, float(err[idx]), float(got[idx]), float(ref[idx]))
        if err[idx] > TOL:
            assert err[idx] <= 30.0 * sens, f"{name}: HIP differs by {err[idx]:.2e} at {idx}, the formulas' own sensitivity is {sens:.2e}"
    rad.close()
    for e, name, idx, sens in sorted(report, reverse=True)[:4]:
        print(f"{workload}: {name} {idx} HIP-vs-oracle {e:.2e}, oracle fma-vs-plain {sens:.2e}")


def test_synthetic_spartacus_single_precision_columns(oracle_lib):
    """BASELINE.json configs[4] at its synthetic shape: ecCKD-32, SPARTACUS with 3-D effects, single precision.  The
    reference's formulation is unstable in single precision (it warns, radiation_config.F90:1144) and which columns go wrong
    depends on the last bit, so -- as in bench.py: check_parity_single -- the HIP path and the oracle's float build are both
    measured against the oracle in DOUBLE precision on the same 2 048 columns, and the HIP path must be at least as close
    to double as the oracle's own float build is (per field: columns off by more than 2e-3, median difference)."""
    import torch
    from bench import check_parity_single
    from ecrad_amd.device import DeviceCase
    ncol = 16384
    config, clear_sky, desc = build_config("spartacus_ecckd32_sp")
    assert config.i_precision == 1 and config.do_3d_effects
    rad = Radiation(config, backend="hip")
    inputs = make_columns(config, ncol, clear_sky)
    n, nlev, sl, th, gas, cloud, aer = inputs
    flux = Flux.allocate(config, n, nlev)
    case = DeviceCase(config, n, nlev, sl, th, gas, cloud, aer, flux)
    st = rad.lib.ecrad_hip_radiation(rad.handle, n, nlev, 1, n, C.byref(case.inputs), C.byref(case.flux))
    assert st == 0, rad.lib.ecrad_hip_last_error(rad.handle).decode()
    rad.lib.ecrad_hip_synchronize(rad.handle)
    torch.cuda.synchronize()

    class TimedBatch:      # what check_parity_single looks at of a bench.py Workload
        pass
    w = TimedBatch()
    w.config, w.ncol, w.case = config, n, case
    res = check_parity_single(w, first_columns(inputs, NCHECK))
    rad.close()
    print("spartacus_ecckd32_sp:", {k: res[k] for k in ("ok", "fields_checked", "fields_failed", "most_columns_off")})
    assert res["columns_checked"] == NCHECK and res["fields_checked"] >= 20
    assert res["ok"], res


def test_single_precision_spartacus_non_finite_columns_of_the_whole_workload(oracle_lib):
    """BASELINE configs[4] at the bench's 100 000 columns, the policy for non-finite fluxes in SINGLE precision.  The reference's
    single-precision SPARTACUS is unstable and says so (radiation_config.F90:1144-1148); its arithmetic -- the oracle's float build --
    returns NaN for about 20 of these columns: one family of profile (a deck of 15 overcast layers over ~55 layers of 1-20 % cloud),
    shortwave only, g-point 17 of the 32-term model (0-based 16, strongly absorbing), where the adding method of section 4.1
    (radiation_spartacus_sw.F90:936-1000) leaves [0, 1] some layers below the deck, a negative albedo meets the 1e-8 floor of
    `top_albedo` in step_migrations (:1606-1721) and the migration distance diverges (profiles/NOTES_r06.md section 4 has the trace).
    The HIP path holds the two albedo matrices to their physical range in its float instantiation (kernel_spartacus.hip) and must
      * return NO non-finite value anywhere (round 5 without the guard: 7 columns),
      * never more such columns than the oracle's float build,
      * stay finite exactly where the oracle's float build is not, with fluxes in the physical range there,
    and the longwave has no non-finite value on either side."""
    import copy
    from bench import oracle_flux_of
    config, clear_sky, desc = build_config("spartacus_ecckd32_sp")
    assert config.i_precision == 1
    inputs = make_columns(config, NCOL, clear_sky)
    n, nlev, sl, th, gas, cloud, aer = inputs
    rad = Radiation(config, backend="hip")
    flux = Flux.allocate(config, n, nlev)
    objs = [copy.deepcopy(o) for o in (sl, th, gas, cloud, aer)]
    rad.radiation(n, nlev, 1, n, *objs, flux)
    rad.close()
    osp = oracle_flux_of(config, copy.deepcopy(inputs))

    def bad_columns(fl, prefix=""):
        bad = np.zeros(n, bool)
        for name, a in fl.arrays.items():
            if not name.startswith(prefix):
                continue
            nf = ~np.isfinite(a)
            bad |= nf.reshape(-1, n).any(axis=0) if a.shape[-1] == n else nf.reshape(n, -1).any(axis=1)
        return np.flatnonzero(bad)
    bad_hip, bad_ora = bad_columns(flux), bad_columns(osp)
    print(f"spartacus_ecckd32_sp, {n} columns: non-finite in {len(bad_hip)} HIP columns, {len(bad_ora)} columns of the oracle's float build "
          f"{bad_ora.tolist()}")
    if len(bad_ora):
        gbad = sorted({int(g) for c in bad_ora for g in np.flatnonzero(~np.isfinite(osp.arrays["sw_up_toa_g"][c]))})
        ncloudy = [int((cloud.fraction[:, c] > 0).sum()) for c in bad_ora]
        print(f"  oracle float: g-points (0-based) {gbad}; cloudy layers per column {min(ncloudy)} .. {max(ncloudy)}")
        # (on the boxes of round 6: 67 .. 73 -- the deep partly cloudy family; not asserted: which columns the float build loses depends on the
        #  host's math library to the last bit)
    assert len(bad_columns(flux, "lw_")) == 0 and len(bad_columns(osp, "lw_")) == 0
    assert len(bad_hip) <= len(bad_ora)
    assert len(bad_hip) == 0, bad_hip.tolist()
    # where the reference's arithmetic gives up the HIP path returns fluxes, and they are fluxes: within [0, incoming] in every such column
    for c in bad_ora:
        toa = flux.arrays["sw_dn"][0, c]
        assert np.all(flux.arrays["sw_up"][:, c] >= -1e-3) and np.all(flux.arrays["sw_up"][:, c] <= toa * (1 + 1e-3) + 1e-3), c
        assert np.all(flux.arrays["sw_dn"][:, c] >= -1e-3) and np.all(flux.arrays["sw_dn"][:, c] <= toa * (1 + 1e-3) + 1e-3), c
