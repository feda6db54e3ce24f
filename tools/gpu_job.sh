#!/bin/bash
# one GPU call: taumol ablations (kernel stats per variant), then the profile of the headline workload
mkdir -p gpurun_out
for v in cur abl16 abl32; do
  echo "== $v"
  ECRAD_HIP_LIB=$PWD/build_variants/$v/libecrad_hip.so bash tools/kstats.sh mcica_rrtmg 2>&1 | head -8
done 2>&1 | tee gpurun_out/r02_f_taumol_ablation.log
bash tools/profile.sh r02_f > gpurun_out/r02_f_profile.log 2>&1
tail -3 gpurun_out/r02_f_profile.log
