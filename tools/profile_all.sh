#!/bin/bash
# tools/profile_all.sh TAG -- on the GPU box: tools/profile.sh of the default bench run (TAG), then one rocprofv3 stats +
# FETCH_SIZE + WRITE_SIZE set per non-headline workload of the default run (TAG_<workload>), each in its own passes as
# the MI355X guide prescribes; summarise with tools/summarize_prof.py TAG [TAG_<workload> ...] afterwards.
TAG=${1:-prof}
export TMPDIR=/tmp
tools/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
for w in tripleclouds_ecckd32 mcica_ecckd32 mcica_rrtmg tripleclouds_ecckd64 spartacus_ecckd32_sp; do
  OUT=$PWD/gpurun_out/${TAG}_$w
  mkdir -p $OUT
  NCOL=100000; [ $w = tripleclouds_ecckd64 ] && NCOL=1250000      # the column counts of the default run (bench.py: EXTRA_WORKLOADS)
  ARGS="--steps 3 --warmup 1 --no-cpu-baseline --headline-only --no-host-mode --workload $w --ncol $NCOL"
  rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python bench.py $ARGS > $OUT/bench_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python bench.py $ARGS > $OUT/bench_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- python bench.py $ARGS > $OUT/bench_write.log 2>&1
  # (the counter calibration is a property of the box and the access width: shared with the default run's)
  cp -r gpurun_out/$TAG/cal_fetch gpurun_out/$TAG/cal_write $OUT/
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only --no-host-mode --workload $w --ncol $NCOL > $OUT/bench.json 2> $OUT/bench.err
  tail -c 600 $OUT/bench.json
done
