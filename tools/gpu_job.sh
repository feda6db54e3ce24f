#!/bin/bash
out=gpurun_out/r04_ah; mkdir -p $out
export TMPDIR=/tmp
timeout 900 bash tools/run_variants.sh --headline-only --workload mcica_ecckd32 < /dev/null 2>&1 | tee $out/variants_mcica.log
