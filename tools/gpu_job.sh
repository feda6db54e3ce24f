#!/bin/bash
# one GPU call: work-budget experiment (one tile instead of two), then the profiles of every workload of the default run
mkdir -p gpurun_out
run() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only --workload $1 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); st = d['roofline']['stage_ms']
        print('%-10s %-22s %10.0f col/s  prep %7.2f lw %7.2f  sw %7.2f tiles %d' % ('$2', '$1', d['value'], st['prep'], st['lw'], st['sw'], d['roofline']['column_tiles']))
"; }
for w in mcica_rrtmg spartacus_ecckd32_sp; do
  run $w 64GiB
  ECRAD_HIP_WORK_GIB=200 run $w 200GiB
done 2>&1 | tee gpurun_out/r02_t_workgib.log
tools/profile_all.sh r02_t
