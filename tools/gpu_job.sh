#!/bin/bash
# one GPU call (diagnostic): what taking columns out of their memory order costs the 32-lane kernels
mkdir -p gpurun_out
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-16s %-26s %10.0f col/s  lw %7.3f  sw %7.3f' % ('$2', '$1', d['value'], st['lw'], st['sw']))
"; }
for w in tripleclouds_ecckd32 mcica_ecckd32; do
ECRAD_SYNTH_SAME_PROFILE=2 ECRAD_NO_COLUMN_ORDER=1 run $w same2_asis
ECRAD_SYNTH_SAME_PROFILE=2 ECRAD_ORDER_WINDOW=16 run $w same2_win16
ECRAD_SYNTH_SAME_PROFILE=2 ECRAD_ORDER_WINDOW=64 run $w same2_win64
ECRAD_SYNTH_SAME_PROFILE=2 ECRAD_ORDER_WINDOW=256 run $w same2_win256
done
