#!/bin/bash
export TMPDIR=/tmp
python tools/sp_column.py 65490 ecrad_amd/csrc/libecrad_hip.so tests/_build/variants/nopack/libecrad_hip.so 2>&1 | grep -v Warning | tee gpurun_out/sp_column.log
for w in clear_homogeneous_ecckd32 tripleclouds_ecckd32 mcica_ecckd32; do
  echo "== $w" | tee -a gpurun_out/r06_cached_top.log
  bash tools/run_variants.sh --headline-only --no-host-mode --workload $w 2>&1 | tee -a gpurun_out/r06_cached_top.log
done
