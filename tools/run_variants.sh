#!/bin/bash
# Run bench.py for every build_variants/*/libecrad_hip.so on the GPU box; prints name, columns/s, LW ms, SW ms.
# ECRAD_VARIANT_PASSES (default 2) interleaved passes over the variants: the box drifts by a few per cent within a call.
passes=${ECRAD_VARIANT_PASSES:-2}
for pass in $(seq 1 $passes); do
for lib in build_variants/*/libecrad_hip.so; do
  n=$(basename $(dirname $lib))
  ECRAD_HIP_LIB=$PWD/$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-28s %10.0f col/s  %8.3f ms  lw %7.3f  sw %7.3f  prep %6.3f' % ('$n', d['value'], d['ms_per_step'], st['lw'], st['sw'], st['prep']))
"
done
done
