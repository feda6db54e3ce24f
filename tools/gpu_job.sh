#!/bin/bash
# one GPU call: cacheable scratch records for the top K levels of the shortwave ICA kernel (written last, read first)
mkdir -p gpurun_out
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-26s %10.0f col/s  lw %7.3f  sw %7.3f' % ('$2', '$1', d['value'], st['lw'], st['sw']))
"; }
for rep in 1 2; do
for w in clear_homogeneous_ecckd32 mcica_ecckd32; do
  run $w current
  for lib in build_variants/*/libecrad_hip.so; do
    ECRAD_HIP_LIB=$PWD/$lib run $w $(basename $(dirname $lib))
  done
done
done
