#!/bin/bash
# tools/host_sweep.sh -- the pipelined host-memory call under /opt/rocm's runtime (no torch in the process): tile size and copy threads
export TMPDIR=/tmp
out=gpurun_out/host_sweep.log; : > $out
W=${1:-clear_homogeneous_ecckd32}
for tile in 8192 12288 16384 24576; do
  for thr in 1,1 2,1 2,2; do
    echo "== tile $tile threads $thr" >> $out
    ECRAD_HIP_HOST_TILE=$tile ECRAD_HIP_COPY_THREADS=$thr python tools/host_link_probe.py notorch $W 100000 2>&1 | grep "arrays" >> $out
  done
done
echo "== no ramp" >> $out
ECRAD_HIP_NO_RAMP=1 python tools/host_link_probe.py notorch $W 100000 2>&1 | grep "arrays" >> $out
cat $out
