#!/bin/bash
# Run bench.py for every build_variants/*/libecrad_hip.so on the GPU box; prints name, columns/s, LW ms, SW ms
for lib in build_variants/*/libecrad_hip.so; do
  n=$(basename $(dirname $lib))
  ECRAD_HIP_LIB=$PWD/$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-28s %10.0f col/s  lw %7.3f  sw %7.3f' % ('$n', d['value'], st['lw'], st['sw']))
"
done
