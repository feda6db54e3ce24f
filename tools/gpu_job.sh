#!/bin/bash
out=gpurun_out/r04_bp; mkdir -p $out
export TMPDIR=/tmp
( ECRAD_HIP_LIB=$PWD/build_variants/b1024t256/libecrad_hip.so timeout 600 python -m pytest tests/test_hip_rrtmg.py -q -m gpu -x < /dev/null ) 2>&1 | tail -3 | tee $out/tests.log
for w in mcica_rrtmg; do
echo "== $w"
ECRAD_VARIANT_PASSES=3 bash tools/run_variants.sh --workload $w --steps 6 --headline-only --no-host-mode 2>&1
done | tee $out/variants.log
