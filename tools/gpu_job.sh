#!/bin/bash
# r03_b: full GPU suite after the parity / SPARTACUS-stage changes; SPARTACUS bench; taumol ablations
mkdir -p gpurun_out/r03_b
O=gpurun_out/r03_b
( time python -m pytest tests -m gpu -x -q ) > $O/tests.log 2>&1; tail -5 $O/tests.log
python bench.py --steps 5 --warmup 1 --workload spartacus_ecckd32_sp --headline-only > $O/bench_sp.json 2> $O/bench_sp.err
python bench.py --steps 3 --warmup 1 --workload spartacus_ecckd32_sp --ncol 1250000 --headline-only --no-cpu-baseline > $O/bench_sp_1250k.json 2> $O/bench_sp_1250k.err
python bench.py --steps 5 --warmup 1 --workload mcica_rrtmg --headline-only --no-cpu-baseline > $O/bench_rrtmg.json 2> $O/bench_rrtmg.err
for v in tau_nostore tau_noeval; do
  ECRAD_HIP_LIB=$PWD/build_variants/$v/libecrad_hip.so python bench.py --steps 5 --warmup 1 --workload mcica_rrtmg --headline-only --no-cpu-baseline > $O/bench_rrtmg_$v.json 2> $O/bench_rrtmg_$v.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_b/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f.split("/")[-1], d["value"], round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()}, d.get("parity"), d["roofline"]["work_bytes"] / 1e9)
    except Exception as e:
        print(f, "failed", e)
PY
