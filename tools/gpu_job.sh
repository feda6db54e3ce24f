#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "tripleclouds or Tripleclouds or tc or golden or targets" 2>&1 | tail -4
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-26s %10.0f col/s  lw %7.3f  sw %7.3f' % ('$2', '$1', d['value'], st['lw'], st['sw']))
"; }
for rep in 1 2; do run tripleclouds_ecckd32 new; run tripleclouds_rrtmg new; done
