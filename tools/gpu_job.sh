#!/bin/bash
# one GPU call: McICA tests, then the McICA workloads (generator: level words outside the cloudy span skipped)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "mcica or McICA or golden or synthetic" 2>&1 | tail -4
run() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-22s %10.0f col/s  prep %7.2f lw %7.2f  sw %7.2f' % ('$2', '$1', d['value'], st['prep'], st['lw'], st['sw']))
"; }
for rep in 1 2; do
for w in mcica_rrtmg mcica_ecckd32; do
  run $w new
done
done
tools/kstats.sh mcica_rrtmg --headline-only 2>&1 | grep generator
