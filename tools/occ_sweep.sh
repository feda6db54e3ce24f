#!/bin/bash
# tools/occ_sweep.sh -- stage times of the bench workload at 1..4 resident blocks per CU
p='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["stage_ms"])'
for b in 1 2 3 4; do
  ECRAD_HIP_BLOCKS_PER_CU=$b timeout -s KILL 200 python bench.py --no-cpu-baseline --steps 5 "$@" 2>/dev/null | python -c "$p" blocks_per_cu=$b
done
