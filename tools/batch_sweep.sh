#!/bin/bash
# tools/batch_sweep.sh -- the reference's OpenMP driver + drop-in, blocks of 80: cap on the columns of a batch x batches in flight x threads
export TMPDIR=/tmp
for thr in 16 64; do
  for cap in 4096 2048 1280 800; do
    for slots in 2 3 4; do
      echo "== threads $thr cap $cap slots $slots"
      TRACE_THREADS=$thr ECRAD_HIP_BATCH_COLUMNS=$cap ECRAD_HIP_SMALL_SLOTS=$slots python tools/dropin_blocks.py 40960 2>&1 | grep "columns/s"
    done
  done
done
