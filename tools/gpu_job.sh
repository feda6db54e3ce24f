#!/bin/bash
out=gpurun_out/r04_by; mkdir -p $out
export TMPDIR=/tmp
( ECRAD_HIP_LIB=$PWD/build_variants/dump4/libecrad_hip.so timeout 900 python -m pytest tests/test_hip_spartacus.py -q -m gpu -x < /dev/null ) 2>&1 | tail -3 | tee $out/tests.log
for w in spartacus_ecckd32_sp; do
echo "== $w"
ECRAD_VARIANT_PASSES=2 bash tools/run_variants.sh --workload $w --steps 6 --headline-only --no-host-mode 2>&1
done | tee $out/variants.log
