#!/bin/bash
mkdir -p gpurun_out/r03_zv
O=gpurun_out/r03_zv
run() {
  local out=$O/$1_$3.json
  ECRAD_HIP_LIB=$2 timeout 300 python bench.py --steps 5 --warmup 2 --workload $3 --headline-only --no-cpu-baseline $4 > $out 2> $O/$1_$3.err
  python - "$out" "$1" "$3" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    st = d["roofline"]["stage_ms"]
    print("%-10s %-28s %10.0f col/s  %8.2f ms  lw %7.2f sw %7.2f prep %6.2f" % (sys.argv[2], sys.argv[3], d["value"] or 0, d["ms_per_step"], st["lw"], st["sw"], st["prep"]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
}
BASE=$PWD/ecrad_amd/csrc/libecrad_hip.so
timeout 1200 python -m pytest tests/test_hip_spartacus.py tests/test_hip_parity.py tests/test_driver_outputs.py -x -q -m gpu 2>&1 | tail -3
for w in spartacus_ecckd32_sp spartacus_ecckd32_dp spartacus_ecckd32_sp; do run shipped $BASE $w; done
bash tools/kstats.sh spartacus_ecckd32_sp 2>&1 | head -7
