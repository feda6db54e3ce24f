#!/bin/bash
out=gpurun_out/r04_bi; mkdir -p $out
export TMPDIR=/tmp
for w in mcica_rrtmg; do
echo "== $w"
ECRAD_VARIANT_PASSES=2 bash tools/run_variants.sh --workload $w --steps 6 --headline-only --no-host-mode 2>&1
done | tee $out/variants.log
( ECRAD_HIP_LIB=$PWD/build_variants/levfast/libecrad_hip.so timeout 900 python -m pytest tests/test_hip_rrtmg.py -q -m gpu -x < /dev/null ) 2>&1 | tail -4 | tee $out/tests.log
