#!/bin/bash
out=gpurun_out/r04_k; mkdir -p $out
export TMPDIR=/tmp
for q in 4 8; do for slots in 2 3; do
echo "== slots $slots queues $q"
GPU_MAX_HW_QUEUES=$q ECRAD_HIP_BATCH_TRACE=1 ECRAD_HIP_SMALL_SLOTS=$slots timeout 300 python tools/small_call_latency.py --host --ncol 80 --threads 32 --contexts 8 > $out/trace_${slots}_$q.log 2>&1; grep "batch:" $out/trace_${slots}_$q.log | tail -6; grep columns/s $out/trace_${slots}_$q.log
done; done
