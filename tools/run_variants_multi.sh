#!/bin/bash
# tools/run_variants_multi.sh TAG workload[:headline] ...: every build_variants/*/libecrad_hip.so on each workload ("headline" = the default one), two passes
TAG=${1:-r05_x}; shift
out=gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
: > $out/variants.log
for w in "$@"; do
  echo "== $w" >> $out/variants.log
  if [ "$w" = headline ]; then
    ECRAD_VARIANT_PASSES=2 bash tools/run_variants.sh --headline-only --no-host-mode --steps 10 >> $out/variants.log 2>&1
  else
    ECRAD_VARIANT_PASSES=2 bash tools/run_variants.sh --headline-only --no-host-mode --workload $w --steps 10 >> $out/variants.log 2>&1
  fi
done
cat $out/variants.log
