#!/bin/bash
out=gpurun_out/r04_ae; mkdir -p $out
export TMPDIR=/tmp
timeout 500 bash tools/run_variants.sh --headline-only --workload spartacus_ecckd32_sp < /dev/null 2>&1 | tee $out/variants_sp.log
ECRAD_VARIANT_PASSES=1 timeout 500 bash tools/run_variants.sh --headline-only --workload tripleclouds_ecckd32 < /dev/null 2>&1 | tee $out/variants_tc.log
( time timeout 2400 python -m pytest tests -q -m gpu -x < /dev/null ) 2>&1 | tail -8 | tee $out/tests.log
timeout 600 python -m pytest tests/test_fortran_dropin.py -q -m gpu -x -s -k "own_timer" < /dev/null 2>&1 | grep -E "columns/s|passed|failed" | tee $out/dropin_timer.log
