#!/bin/bash
# one GPU call: full GPU suite, the driver's bench command, and the profile of the RRTMG McICA workload on this build
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r02_zi_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_zi_bench.json 2> gpurun_out/r02_zi_bench.err
w=mcica_rrtmg; OUT=$PWD/gpurun_out/r02_zi_$w; mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --headline-only --workload $w --ncol 100000"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python bench.py $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/cal_fetch -o cal -- tools/hbm_calibrate > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/cal_write -o cal -- tools/hbm_calibrate > $OUT/cal_write.log 2>&1
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only --workload $w > $OUT/bench.json 2> $OUT/bench.err
head -c 300 gpurun_out/r02_zi_bench.json
