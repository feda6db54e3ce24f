#!/bin/bash
out=gpurun_out/r04_bl; mkdir -p $out
export TMPDIR=/tmp
for w in mcica_rrtmg; do
echo "== $w"
ECRAD_VARIANT_PASSES=2 bash tools/run_variants.sh --workload $w --steps 4 --headline-only --no-host-mode 2>&1
echo "== $w without aerosol folding"
ECRAD_NO_AEROSOL_FOLD=1 ECRAD_VARIANT_PASSES=1 bash tools/run_variants.sh --workload $w --steps 4 --headline-only --no-host-mode 2>&1
done | tee $out/variants.log
