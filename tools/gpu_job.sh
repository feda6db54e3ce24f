#!/bin/bash
out=gpurun_out/r04_b; mkdir -p $out
export TMPDIR=/tmp
bash tools/run_variants.sh --headline-only 2>&1 | tee $out/variants.log
# the start-up flake behind the drop-in tests' retry: the 54 cases in a loop without the retry, every failure's output kept
for i in 1 2 3; do
  ECRAD_TEST_NO_RETRY=1 timeout 600 python -m pytest tests/test_fortran_dropin.py -q -m gpu -x 2>&1 | tail -40 > $out/dropin_loop_$i.log
  tail -2 $out/dropin_loop_$i.log
done
