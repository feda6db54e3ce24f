#!/bin/bash
out=gpurun_out/r04_bk; mkdir -p $out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_synthetic_workload.py -q -m gpu -x < /dev/null ) 2>&1 | tail -4 | tee $out/tests.log
for w in clear_homogeneous_ecckd32 mcica_ecckd32 mcica_rrtmg; do
echo "== $w"
ECRAD_VARIANT_PASSES=3 bash tools/run_variants.sh --workload $w --steps 10 --headline-only --no-host-mode 2>&1
done | tee $out/variants.log
