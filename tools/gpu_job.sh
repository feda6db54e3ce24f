#!/bin/bash
out=gpurun_out/r04_bw; mkdir -p $out
export TMPDIR=/tmp
for w in spartacus_ecckd32_sp; do
echo "== $w"
ECRAD_VARIANT_PASSES=2 bash tools/run_variants.sh --workload $w --steps 6 --headline-only --no-host-mode 2>&1
done | tee $out/variants.log
