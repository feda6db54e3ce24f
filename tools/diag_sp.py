"""Single-precision SPARTACUS on the bench columns: who is off where?  HIP (float), the oracle's float build and the oracle in
double on the same columns.  (GPU box)  python tools/diag_sp.py [NCOL]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
w = bench.Workload("spartacus_ecckd32_sp", ncol, 0, 0, 0)
w.step(); torch.cuda.synchronize()
inp = w.host_inputs
osp = bench.oracle_flux_of(w.config, inp)
cfg_dp, _, _ = bench.build_config("spartacus_ecckd32_dp")
odp = bench.oracle_flux_of(cfg_dp, inp)
for name in ("sw_up", "sw_dn", "sw_up_toa_g", "sw_dn_diffuse_surf_g", "sw_up_clear", "lw_up_clear", "lw_up", "lw_dn"):
    hip = w.case.flux_tensors[name].cpu().numpy()
    sp, dp = osp.arrays[name], odp.arrays[name]
    scale = np.maximum(np.abs(dp), 1e-3 * np.abs(dp).max())
    col_last = hip.shape[-1] == ncol
    ax = tuple(range(hip.ndim - 1)) if col_last else tuple(range(1, hip.ndim))
    def colmax(a):
        e = np.abs(a - dp) / scale
        e = np.where(np.isfinite(e), e, np.inf)
        return e.max(axis=ax)
    eh, eo = colmax(hip), colmax(sp)
    ehs = np.where(np.isfinite(np.abs(hip - sp)), np.abs(hip - sp) / scale, np.inf).max(axis=ax)
    for lab, e in (("hip-vs-dp", eh), ("osp-vs-dp", eo), ("hip-vs-osp", ehs)):
        print(f"{name:22s} {lab:11s} median {np.median(e):.2e}  p99 {np.percentile(e, 99):.2e}  p99.9 {np.percentile(e, 99.9):.2e}  "
              f">2e-3: {int((e > 2e-3).sum())}  >1e-1: {int((e > 1e-1).sum())}  nonfinite: {int(np.isinf(e).sum())}")
    both = (eh > 2e-3) & (eo > 2e-3)
    print(f"{'':22s} columns off by >2e-3 in both: {int(both.sum())}, hip only {int(((eh > 2e-3) & ~(eo > 2e-3)).sum())}, oracle-sp only {int((~(eh > 2e-3) & (eo > 2e-3)).sum())}")
