#!/bin/bash
out=gpurun_out/r04_ce; mkdir -p $out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -q -m gpu -x < /dev/null ) 2>&1 | tail -8 | tee $out/tests.log
timeout 1200 python bench.py < /dev/null > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $out/smoke.log
