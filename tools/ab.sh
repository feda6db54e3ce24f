#!/bin/bash
# tools/ab.sh -- A/B the bench line of an older worktree (ab/old) against the current tree on the same box
p='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), d["roofline"]["stage_ms"])'
for rep in 1 2; do
  (cd ab/old && timeout -s KILL 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$p" old)
  timeout -s KILL 200 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "$p" new
done
