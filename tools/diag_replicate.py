import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import make_config, run_case, load_meridian
from test_hip_parity import _replicate
solver = sys.argv[1]; times = int(sys.argv[2])
c = make_config(solver); f32, _, r = run_case(c, "hip"); r.close()
c2 = make_config(solver); fb, _, r2 = run_case(c2, "hip", inputs=_replicate(load_meridian(c2), times)); r2.close()
for name, a in fb.arrays.items():
    b = f32.arrays[name]
    if a.ndim == 1: want = np.concatenate([b]*times)
    elif a.shape[-1] == 32*times: want = np.concatenate([b]*times, axis=-1)
    else: want = np.concatenate([b]*times, axis=0)
    if not np.array_equal(a, want):
        d = np.abs(a - want); sc = np.abs(want).max() + 1e-300
        idx = np.argwhere(d > 0)
        cols = np.unique(idx[:, -1] if a.shape[-1] == 32*times else idx[:, 0])
        print(name, "max abs", d.max(), "rel", d.max()/sc, "ndiff", len(idx), "cols differing", len(cols), "first cols", cols[:10], "col%32", np.unique(cols % 32)[:20])
