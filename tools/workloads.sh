#!/bin/bash
# tools/workloads.sh WORKLOAD... -- bench line summary (columns/s, stage ms) for each named workload
p='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-28s %9d col/s " % (sys.argv[1], d["value"]), {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()})'
for w in "$@"; do
  timeout -s KILL 300 python bench.py --no-cpu-baseline --steps 5 --workload $w 2>/dev/null | python -c "$p" $w
done
