#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "spartacus or stage or optics or mixed" 2>&1 | tail -4
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --headline-only --workload $1 $3 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-26s %10.0f col/s  prep %6.2f lw %7.3f  sw %7.3f' % ('$2', '$1', d['value'], st['prep'], st['lw'], st['sw']))
"; }
run spartacus_ecckd32_sp new
run spartacus_ecckd32_dp new
tools/kstats.sh spartacus_ecckd32_sp --headline-only 2>&1 | head -7
