#!/bin/bash
mkdir -p gpurun_out/r03_g
O=gpurun_out/r03_g
python -m pytest tests/test_fortran_dropin.py -m gpu -x -q -k spartacus > $O/tests.log 2>&1; tail -3 $O/tests.log
for w in spartacus_ecckd32_sp tripleclouds_ecckd32 mcica_rrtmg clear_homogeneous_ecckd32; do
  echo "== $w" >> $O/kstats.log
  bash tools/kstats.sh $w >> $O/kstats.log 2>&1
done
cat $O/kstats.log
