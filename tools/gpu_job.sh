#!/bin/bash
# r03_l: full GPU suite on the current build; IFS-caller test repeated; a few more variants
mkdir -p gpurun_out/r03_l
O=gpurun_out/r03_l
( time python -m pytest tests -m gpu -x -q ) > $O/tests.log 2>&1; tail -6 $O/tests.log
for i in 1 2 3 4 5; do python -m pytest tests/test_fortran_dropin.py -m gpu -q -k second_caller 2>&1 | tail -1; done
run() {
  local out=$O/$1_$3.json
  ECRAD_HIP_LIB=$2 python bench.py --steps 5 --warmup 2 --workload $3 --headline-only --no-cpu-baseline $4 > $out 2> $O/$1_$3.err
  python - "$out" "$1" "$3" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    st = d["roofline"]["stage_ms"]
    print("%-10s %-28s %10.0f col/s  %8.2f ms  lw %7.2f sw %7.2f prep %6.2f" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], st["lw"], st["sw"], st["prep"]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
}
BASE=$PWD/ecrad_amd/csrc/libecrad_hip.so
for w in tripleclouds_ecckd32 mcica_ecckd32 mcica_rrtmg; do run shipped $BASE $w; done
run shipped $BASE tripleclouds_ecckd64 "--ncol 1250000"
run tcb4 $PWD/build_variants/tcb4/libecrad_hip.so tripleclouds_ecckd32
for v in cld3 cld4; do for w in mcica_ecckd32 mcica_rrtmg; do run $v $PWD/build_variants/$v/libecrad_hip.so $w; done; done
run tcmw2 $PWD/build_variants/tcmw2/libecrad_hip.so tripleclouds_ecckd64 "--ncol 1250000"
