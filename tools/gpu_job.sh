#!/bin/bash
mkdir -p gpurun_out/r03_f
O=gpurun_out/r03_f
python -m pytest tests/test_fortran_dropin.py tests/test_fortran_netcdf.py tests/test_fortran_host.py -m gpu -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
python bench.py --steps 5 --warmup 1 --workload spartacus_ecckd32_sp --headline-only > $O/bench_sp.json 2> $O/bench_sp.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_f/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f.split("/")[-1], d["value"], round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()}, json.dumps(d.get("parity"))[:1500])
    except Exception as e:
        print(f, "failed", e)
PY
tail -5 $O/bench_sp.err
