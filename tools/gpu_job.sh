#!/bin/bash
# one GPU call: taumol with 2 g-points per lane vs 1 (build_variants/g1), after the RRTMG tests
mkdir -p gpurun_out
python -m pytest tests/test_hip_rrtmg.py tests/test_mixed_gas.py -m gpu -x -q 2>&1 | tail -4
run() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-22s %10.0f col/s  prep %7.2f lw %7.2f  sw %7.2f' % ('$2', '$1', d['value'], st['prep'], st['lw'], st['sw']))
"; }
for rep in 1 2; do
for w in mcica_rrtmg; do
  run $w g2
  ECRAD_HIP_LIB=$PWD/build_variants/g1/libecrad_hip.so run $w g1
done
done
