#!/bin/bash
# one GPU call: parity tests, then the tuning variants side by side
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_e_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02_e_tests.log
tail -5 gpurun_out/r02_e_tests.log
for w in clear_homogeneous_ecckd32 tripleclouds_ecckd32 mcica_ecckd32 mcica_rrtmg; do
  echo "== $w"; bash tools/run_variants.sh --headline-only --workload $w
done 2>&1 | tee gpurun_out/r02_e_variants.log
