#!/bin/bash
# tools/profile_round.sh TAG -- on the GPU box: the profile set of a round (tools/profile_all.sh TAG: per workload rocprofv3
# kernel stats + FETCH_SIZE + WRITE_SIZE passes + an un-profiled bench line), the SQ counters of the headline and Tripleclouds
# workloads (tools/pmc_sq.sh), summarised HERE (tools/summarize_prof.py -> profiles/TAG*.md, *_traffic.json) and copied to
# gpurun_out/TAG_summaries/; the raw rocprofv3 databases (hundreds of MB) are deleted: gpurun merges back at most 64 MiB.
TAG=${1:-prof}
export TMPDIR=/tmp
out=gpurun_out/${TAG}_summaries; mkdir -p $out
timeout 3300 bash tools/profile_all.sh $TAG < /dev/null > $out/profile_all.log 2>&1
for t in $TAG ${TAG}_tripleclouds_ecckd32 ${TAG}_mcica_ecckd32 ${TAG}_mcica_rrtmg ${TAG}_tripleclouds_ecckd64 ${TAG}_spartacus_ecckd32_sp; do
  timeout 300 python tools/summarize_prof.py $t < /dev/null > /dev/null 2>> $out/summarize.err
  cp profiles/$t.md profiles/${t}_traffic.json $out/ 2>/dev/null
done
cp gpurun_out/$TAG/bench.json $out/${TAG}_bench.json 2>/dev/null
timeout 500 bash tools/pmc_sq.sh ${TAG}_sq --headline-only < /dev/null > $out/sq_headline.log 2>&1
timeout 500 bash tools/pmc_sq.sh ${TAG}_sq_tc --headline-only --workload tripleclouds_ecckd32 < /dev/null > $out/sq_tripleclouds.log 2>&1
timeout 500 bash tools/pmc_sq.sh ${TAG}_sq_mc --headline-only --workload mcica_ecckd32 < /dev/null > $out/sq_mcica.log 2>&1
timeout 600 bash tools/pmc_sq.sh ${TAG}_sq_rr --headline-only --workload mcica_rrtmg < /dev/null > $out/sq_rrtmg.log 2>&1
timeout 600 bash tools/pmc_sq.sh ${TAG}_sq_sp --headline-only --workload spartacus_ecckd32_sp < /dev/null > $out/sq_spartacus.log 2>&1
python tools/summarize_sq.py $TAG $out > $out/sq_summary.txt 2>&1; cp profiles/${TAG}_sq.json $out/ 2>/dev/null
find gpurun_out -name "*.db" -delete
rm -rf gpurun_out/$TAG/stats gpurun_out/$TAG/fetch gpurun_out/$TAG/write
du -sh gpurun_out | tail -1
ls $out
