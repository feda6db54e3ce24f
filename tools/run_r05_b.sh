#!/bin/bash
# round 5, call b: ablations of the Tripleclouds kernels + the early aerosol-weight fetch, on tripleclouds_ecckd32 and mcica_ecckd32
out=gpurun_out/r05_b; mkdir -p $out
export TMPDIR=/tmp
echo "== tripleclouds_ecckd32" > $out/variants.log
ECRAD_VARIANT_PASSES=2 bash tools/run_variants.sh --headline-only --no-host-mode --workload tripleclouds_ecckd32 --steps 10 >> $out/variants.log 2>&1
echo "== mcica_ecckd32" >> $out/variants.log
mkdir -p /tmp/hold && mv build_variants/abl* build_variants/noahead /tmp/hold/
ECRAD_VARIANT_PASSES=2 bash tools/run_variants.sh --headline-only --no-host-mode --workload mcica_ecckd32 --steps 10 >> $out/variants.log 2>&1
echo "== headline" >> $out/variants.log
ECRAD_VARIANT_PASSES=1 bash tools/run_variants.sh --headline-only --no-host-mode --steps 10 >> $out/variants.log 2>&1
cat $out/variants.log
python -m pytest tests/test_hip_parity.py tests/test_synthetic_workload.py -q -m gpu -x 2>&1 | tail -3
