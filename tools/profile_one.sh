#!/bin/bash
# tools/profile_one.sh TAG WORKLOAD [NCOL] -- on the GPU box: the profile set of ONE non-headline workload (what tools/profile_all.sh does per workload:
# rocprofv3 kernel stats + FETCH_SIZE + WRITE_SIZE passes + the counter calibration + an un-profiled bench line), summarised into profiles/TAG_WORKLOAD.md
# and _traffic.json and copied to gpurun_out/TAG_summaries/.
TAG=${1:-prof}; w=${2:-mcica_rrtmg}; NCOL=${3:-100000}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${TAG}_$w; mkdir -p $OUT gpurun_out/${TAG}_summaries
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --headline-only --no-host-mode --workload $w --ncol $NCOL"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python bench.py $ARGS > $OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python bench.py $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python bench.py $ARGS > $OUT/bench_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/cal_fetch -o cal -- tools/hbm_calibrate > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/cal_write -o cal -- tools/hbm_calibrate > $OUT/cal_write.log 2>&1
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only --no-host-mode --workload $w --ncol $NCOL > $OUT/bench.json 2> $OUT/bench.err
python tools/summarize_prof.py ${TAG}_$w > /dev/null 2> gpurun_out/${TAG}_summaries/summarize_$w.err
cp profiles/${TAG}_$w.md profiles/${TAG}_${w}_traffic.json gpurun_out/${TAG}_summaries/ 2>/dev/null
timeout 600 bash tools/pmc_sq.sh ${TAG}_sq_one --headline-only --workload $w < /dev/null > gpurun_out/${TAG}_summaries/sq_$w.log 2>&1
find gpurun_out -name "*.db" -delete; rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/cal_fetch $OUT/cal_write
ls gpurun_out/${TAG}_summaries | head -20; tail -c 400 $OUT/bench.json
