"""Where are the non-finite values of a bench workload?  (GPU box)  python tools/diag_nan.py WORKLOAD [NCOL]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "spartacus_ecckd32_sp"
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
w = bench.Workload(name, ncol, 0, 0, 0)
w.step(); torch.cuda.synchronize()
bad = np.zeros(ncol, dtype=bool)
for n, t in w.case.flux_tensors.items():
    a = t.cpu().numpy()
    nf = ~np.isfinite(a)
    if nf.any():
        cols = np.unique(np.nonzero(nf)[-1] if a.shape[-1] == ncol else np.nonzero(nf)[0])
        print(n, "non-finite values:", int(nf.sum()), "columns:", cols[:20], "..." if len(cols) > 20 else "")
        bad[cols] = True
cols = np.nonzero(bad)[0]
print("columns with non-finite values:", len(cols), cols[:50])
if len(cols):
    inp = w.host_inputs
    ncol_, nlev, sl, th, gas, cloud, aer = inp
    for c in cols[:5]:
        f = cloud.fraction[:, c]
        print("column", c, "mu0", sl.cos_sza[c], "cloudy layers", np.nonzero(f > 0)[0], "fractions", f[f > 0][:10])
        blk = bench.columns_of(inp, (c // 32) * 32, 32)
        of = bench.oracle_flux_of(w.config, blk)
        j = c - (c // 32) * 32
        for n in ("sw_up", "sw_dn", "lw_up"):
            print("  oracle", n, of.arrays[n][:3, j], " hip", w.case.flux_tensors[n][:3, c].cpu().numpy())
