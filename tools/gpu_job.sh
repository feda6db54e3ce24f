#!/bin/bash
mkdir -p gpurun_out/r03_n
O=gpurun_out/r03_n
python -m pytest tests/test_fortran_dropin.py -m gpu -q -s 2>&1 | grep -E "max|passed|failed|single-precision" > $O/dropin.log; tail -22 $O/dropin.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.log; tail -3 $O/bench_time.log; tail -c 300 $O/bench_default.json
