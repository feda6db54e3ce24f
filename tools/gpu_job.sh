#!/bin/bash
# one GPU call: McICA generators on a second stream -- tests, then A/B against ECRAD_NO_GEN_OVERLAP
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "mcica or McICA or golden or synthetic or tiled or mixed or order" 2>&1 | tail -4
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-26s %10.0f col/s  %7.2f ms  prep %6.2f lw %7.3f  sw %7.3f' % ('$2', '$1', d['value'], d['ms_per_step'], st['prep'], st['lw'], st['sw']))
"; }
for rep in 1 2; do
for w in mcica_ecckd32 mcica_rrtmg; do
  ECRAD_NO_GEN_OVERLAP=1 run $w serial
  run $w overlap
done
done
