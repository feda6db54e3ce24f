#!/bin/bash
out=gpurun_out/r04_ai; mkdir -p $out
export TMPDIR=/tmp
for share in 0 1; do for slots in 2 3 4; do
  echo "== ECRAD_HIP_BATCH_SHARE=$share ECRAD_HIP_SMALL_SLOTS=$slots"
  ECRAD_HIP_BATCH_SHARE=$share ECRAD_HIP_SMALL_SLOTS=$slots timeout 300 python tools/small_call_latency.py --host --ncol 80 --threads 16,32 --contexts 8 --solver Tripleclouds < /dev/null 2>&1 | grep -E "columns/s" | tail -3
done; done 2>&1 | tee $out/share.log
echo "== McICA share=1 slots=3"; ECRAD_HIP_BATCH_SHARE=1 ECRAD_HIP_SMALL_SLOTS=3 timeout 300 python tools/small_call_latency.py --host --ncol 80 --threads 16,32 --contexts 8 --solver McICA < /dev/null 2>&1 | grep -E "columns/s" | tail -3 | tee -a $out/share.log
timeout 600 python -m pytest tests/test_hip_pool.py -q -m gpu -x < /dev/null 2>&1 | tail -3 | tee $out/tests.log
