#!/bin/bash
out=gpurun_out/r04_r; mkdir -p $out
export TMPDIR=/tmp
bash tools/run_variants.sh --headline-only --workload spartacus_ecckd32_sp 2>&1 | tee $out/variants_sp.log
ECRAD_VARIANT_PASSES=1 bash tools/run_variants.sh --headline-only --workload spartacus_ecckd32_dp 2>&1 | tee $out/variants_dp.log
timeout 900 python -m pytest tests/test_hip_spartacus.py tests/test_reference_suites.py tests/test_hip_rrtmg.py -q -m gpu -x -s 2>&1 | grep -E "single precision, do_3d|passed|failed|FAIL|Error" | tee $out/tests.log
