#!/bin/bash
# one GPU call: the cloud generator with two 64-level words where the clouds allow it -- McICA tests, then the McICA workloads
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "mcica or McICA or golden or synthetic or tiled or mixed" 2>&1 | tail -4
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-26s %10.0f col/s  prep %6.2f lw %7.3f  sw %7.3f' % ('$2', '$1', d['value'], st['prep'], st['lw'], st['sw']))
"; }
for w in mcica_ecckd32 mcica_rrtmg; do run $w gen1w; done
tools/kstats.sh mcica_rrtmg --headline-only 2>&1 | grep generator
