#!/bin/bash
out=gpurun_out/r04_u; mkdir -p $out
export TMPDIR=/tmp
timeout 400 bash tools/run_variants.sh --headline-only < /dev/null 2>&1 | tee $out/variants.log
for w in tripleclouds_ecckd32 mcica_ecckd32 spartacus_ecckd32_sp; do
  echo "== $w"; ECRAD_VARIANT_PASSES=1 timeout 400 bash tools/run_variants.sh --headline-only --workload $w < /dev/null 2>&1
done | tee $out/variants_other.log
timeout 300 python bench.py --steps 5 --no-cpu-baseline --headline-only < /dev/null 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(json.dumps(d.get('end_to_end_host'), indent=1)); print(d['value'], d.get('parity'))
" | tee $out/host.log
timeout 600 python -m pytest tests/test_hip_parity.py -q -m gpu -x -k "compile_time or packed or golden or stage" < /dev/null 2>&1 | tail -5 | tee $out/tests.log
