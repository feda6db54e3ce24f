#!/bin/bash
# one GPU call: ingredient variants of the level-record phase on the cloudy workloads
mkdir -p gpurun_out
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-14s %-26s %10.0f col/s  lw %7.3f  sw %7.3f' % ('$2', '$1', d['value'], st['lw'], st['sw']))
"; }
for rep in 1 2; do
for w in tripleclouds_ecckd32 mcica_ecckd32; do
  run $w current
  for lib in build_variants/*/libecrad_hip.so; do
    ECRAD_HIP_LIB=$PWD/$lib run $w $(basename $(dirname $lib))
  done
done
done 2>&1 | tee gpurun_out/r02_k_variants.log
