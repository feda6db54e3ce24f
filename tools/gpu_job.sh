#!/bin/bash
out=gpurun_out/r04_bs; mkdir -p $out
export TMPDIR=/tmp
p='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-40s %9d col/s %8.2f ms " % (sys.argv[1], d["value"], d["ms_per_step"]), {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()})'
run() { timeout -s KILL 300 python bench.py --no-cpu-baseline --steps 6 --headline-only --no-host-mode --workload mcica_rrtmg 2>/dev/null | python -c "$p" "$1"; }
for rep in 1 2 3; do
run default
ECRAD_GEN_OVERLAP=1 run gen_overlap
ECRAD_GEN_SW_LATE=1 run gen_sw_late
done 2>&1 | tee $out/knobs.log
timeout 1200 python bench.py < /dev/null > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json
