#!/bin/bash
# tools/profile.sh TAG [bench args...] -- run on the GPU box (via gpurun).  Collects, under gpurun_out/TAG/:
#   stats/   rocprofv3 --kernel-trace --stats of `python bench.py ...`
#   fetch/   rocprofv3 --pmc FETCH_SIZE   (own pass, as the MI355X guide prescribes)
#   write/   rocprofv3 --pmc WRITE_SIZE   (own pass)
#   cal_*/   the same two PMC passes over tools/hbm_calibrate (known byte counts, same 8 B/lane width)
TAG=${1:-prof}; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --headline-only --no-host-mode $@"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python bench.py $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/cal_fetch -o cal -- tools/hbm_calibrate > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/cal_write -o cal -- tools/hbm_calibrate > $OUT/cal_write.log 2>&1
python bench.py --steps 10 --warmup 2 "$@" > $OUT/bench.json 2> $OUT/bench.err
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
find $OUT -name "*.csv" | head -50
tail -2 $OUT/bench.json
