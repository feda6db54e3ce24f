#!/bin/bash
out=gpurun_out/r04_bx; mkdir -p $out
export TMPDIR=/tmp
p='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-28s %9d col/s %8.2f ms " % (sys.argv[1], d["value"], d["ms_per_step"]), {k: round(v, 2) for k, v in d["roofline"]["stage_ms"].items()})'
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --steps 6 --headline-only --no-host-mode --workload tripleclouds_ecckd32 2>/dev/null | python -c "$p" a_new
ECRAD_HIP_BLOCKS_PER_CU=2 ECRAD_HIP_LIB=$PWD/build_variants/tc2/libecrad_hip.so timeout 300 python bench.py --no-cpu-baseline --steps 6 --headline-only --no-host-mode --workload tripleclouds_ecckd32 2>/dev/null | python -c "$p" tc2_blocks2
done 2>&1 | tee $out/tc2.log
( time timeout 2400 python -m pytest tests -q -m gpu -x < /dev/null ) 2>&1 | tail -8 | tee $out/tests.log
timeout 1200 python bench.py < /dev/null > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $out/smoke.log
