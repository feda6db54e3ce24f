#!/bin/bash
out=gpurun_out/r04_cd; mkdir -p $out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_hip_spartacus.py tests/test_reference_suites.py tests/test_synthetic_workload.py -q -m gpu -x < /dev/null ) 2>&1 | tail -4 | tee $out/tests.log
bash tools/workloads.sh spartacus_ecckd32_dp 2>&1 | tee $out/workloads.log
