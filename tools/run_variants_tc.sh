#!/bin/bash
# tools/run_variants_tc.sh TAG [workloads...]: every build_variants/*/libecrad_hip.so on the named workloads (default: tripleclouds_ecckd32), two passes
TAG=${1:-r05_x}; shift
out=gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
W="$@"; [ -z "$W" ] && W="tripleclouds_ecckd32"
: > $out/variants.log
for w in $W; do
  echo "== $w" >> $out/variants.log
  ECRAD_VARIANT_PASSES=2 bash tools/run_variants.sh --headline-only --no-host-mode --workload $w --steps 10 >> $out/variants.log 2>&1
done
cat $out/variants.log
