#!/bin/bash
# r03_a: new launcher / packing tests, headline parity over all 100 000 columns with and without contraction in the classic two-stream routine
mkdir -p gpurun_out/r03_a
O=gpurun_out/r03_a
python -m pytest tests/test_bench_launcher.py tests/test_hip_parity.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
python bench.py --steps 10 --warmup 2 --headline-only > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
ECRAD_HIP_LIB=$PWD/build_variants/contract/libecrad_hip.so python bench.py --steps 10 --warmup 2 --headline-only > $O/bench_contract.json 2> $O/bench_contract.err
python - <<'PY'
import json
for n in ("default", "contract"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r03_a/bench_{n}.json") if l.startswith("{")][0])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["stage_ms"], d["parity"], d["roofline"].get("measured_triad"), d["roofline"].get("fp64_fraction"))
    except Exception as e:
        print(n, "failed", e)
PY
