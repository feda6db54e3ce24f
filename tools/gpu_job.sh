#!/bin/bash
# one GPU call: chunk plan A/B (chunks of different widths vs one width) on the wide-spectrum workloads, after the tests that cover it
mkdir -p gpurun_out
true
run() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-22s %10.0f col/s  prep %7.2f lw %7.2f  sw %7.2f  parity %.2e' % ('$2', '$1', d['value'], st['prep'], st['lw'], st['sw'], (d.get('parity') or {}).get('max_rel_diff_vs_oracle', float('nan'))))
"; }
for rep in 1 2; do
for w in mcica_rrtmg tripleclouds_rrtmg; do
  run $w mixed
  ECRAD_CHUNK_PLAN=uniform run $w uniform
done
done
