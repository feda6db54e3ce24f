#!/bin/bash
# tools/chunk_sweep.sh WORKLOAD -- bench line of a wide-spectrum workload for each forced chunk width (ECRAD_CHUNK_LANES)
p='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["stage_ms"])'
for n in 16 32 64; do
  ECRAD_CHUNK_LANES=$n timeout -s KILL 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --workload $1 2>/dev/null | python -c "$p" lanes=$n
done
