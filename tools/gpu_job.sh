#!/bin/bash
# one GPU call: per-spectrum ordering windows -- full test suite, then every cloudy workload against ECRAD_NO_COLUMN_ORDER
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --headline-only --workload $1 $3 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-26s %10.0f col/s  prep %6.2f lw %7.3f  sw %7.3f' % ('$2', '$1', d['value'], st['prep'], st['lw'], st['sw']))
"; }
for w in tripleclouds_ecckd32 mcica_ecckd32 mcica_rrtmg tripleclouds_rrtmg spartacus_ecckd32_sp; do
  ECRAD_NO_COLUMN_ORDER=1 run $w asis
  run $w ordered
done
ECRAD_NO_COLUMN_ORDER=1 run tripleclouds_ecckd64 asis "--ncol 1250000"
run tripleclouds_ecckd64 ordered "--ncol 1250000"
