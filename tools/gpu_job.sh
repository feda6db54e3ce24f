#!/bin/bash
# r03_h: the drop-in targets, then the profile set of this build: stats + PMC traffic of the headline and of each workload
mkdir -p gpurun_out
python -m pytest tests/test_fortran_dropin.py -m gpu -q > gpurun_out/r03_h_tests.log 2>&1; tail -4 gpurun_out/r03_h_tests.log
bash tools/profile_all.sh r03_h > gpurun_out/r03_h_all.log 2>&1
python tools/summarize_prof.py r03_h > gpurun_out/r03_h_summary.log 2>&1
for w in tripleclouds_ecckd32 mcica_rrtmg tripleclouds_ecckd64 spartacus_ecckd32_sp; do python tools/summarize_prof.py r03_h_$w >> gpurun_out/r03_h_summary.log 2>&1; done
tail -5 gpurun_out/r03_h_summary.log; ls profiles | grep r03_h
cp profiles/r03_h* gpurun_out/ 2>/dev/null
tail -c 3000 gpurun_out/r03_h/bench.json
