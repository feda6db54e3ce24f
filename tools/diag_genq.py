"""(GPU box) McICA generator with the column queue against the oracle, column by column."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from helpers import make_config, run_case
from oracle import pyoracle
pyoracle.build(ref=False)
kw = {}
f_ora, _, _ = run_case(make_config("McICA", None, **kw), pyoracle.backend)
for rep in range(2):
    f_hip, _, rad = run_case(make_config("McICA", None, **kw), "hip")
    for name in ("lw_dn", "sw_dn", "cloud_cover_lw", "cloud_cover_sw"):
        a, b = f_hip.arrays[name], f_ora.arrays[name]
        if a.ndim == 2: a, b = a[-1], b[-1]
        d = np.abs(a - b) / (np.abs(b) + 1e-30)
        print(rep, name, "bad columns:", np.nonzero(d > 1e-8)[0].tolist())
    rad.close()
