import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import numpy as np
from helpers import make_config, load_meridian
from test_hip_tiling import _run, _replicate
def go(env, **kw):
    if env: os.environ["ECRAD_NO_COLUMN_ORDER"] = "1"
    else: os.environ.pop("ECRAD_NO_COLUMN_ORDER", None)
    c = make_config("SPARTACUS", do_3d_effects=True, do_lw_derivatives=True, **kw)
    f, frac, _ = _run(c, _replicate(load_meridian(c), 40))
    return f
for kw in (dict(), dict(do_3d_effects=False)):
    a = go(False, **kw); b = go(True, **kw); c = go(True, **kw); d = go(False, **kw)
    for name in a.arrays:
        x, y, z, w = a.arrays[name], b.arrays[name], c.arrays[name], d.arrays[name]
        if not np.array_equal(x, y) or not np.array_equal(y, z) or not np.array_equal(x, w):
            dif = np.abs(x - y); i = np.unravel_index(np.argmax(dif), dif.shape)
            print(kw, name, "ordered vs asis max", dif.max(), "at", i, "rel", dif.max() / (np.abs(y).max() + 1e-300),
                  "| asis vs asis equal:", np.array_equal(y, z), "| ordered vs ordered equal:", np.array_equal(x, w),
                  "| n differing columns", int((dif.reshape(dif.shape[0], -1) if dif.ndim > 1 else dif[None]).any(axis=0).sum()) if dif.ndim > 1 else int((dif > 0).sum()))
print("done")
