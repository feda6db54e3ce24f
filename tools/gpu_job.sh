#!/bin/bash
out=gpurun_out/r04_x; mkdir -p $out
export TMPDIR=/tmp
for ct in 1,1 2,2 3,3 4,4 2,4 4,2; do
  echo "== ECRAD_HIP_COPY_THREADS=$ct"
  ECRAD_HIP_COPY_THREADS=$ct timeout 200 python tools/host_mode_rate.py clear_homogeneous_ecckd32 100000 < /dev/null 2>&1 | tail -3
done 2>&1 | tee $out/host_threads.log
for tile in 6144 8192 16384 24576; do
  echo "== ECRAD_HIP_HOST_TILE=$tile (threads 3,3)"
  ECRAD_HIP_COPY_THREADS=3,3 ECRAD_HIP_HOST_TILE=$tile timeout 200 python tools/host_mode_rate.py clear_homogeneous_ecckd32 100000 < /dev/null 2>&1 | tail -2
done 2>&1 | tee $out/host_tiles.log
echo "== tripleclouds 3,3"; ECRAD_HIP_COPY_THREADS=3,3 timeout 200 python tools/host_mode_rate.py tripleclouds_ecckd32 100000 < /dev/null 2>&1 | tail -2 | tee $out/host_tc.log
echo "== tripleclouds 2,2"; ECRAD_HIP_COPY_THREADS=2,2 timeout 200 python tools/host_mode_rate.py tripleclouds_ecckd32 100000 < /dev/null 2>&1 | tail -2 | tee -a $out/host_tc.log
