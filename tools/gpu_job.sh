#!/bin/bash
out=gpurun_out/r04_cb; mkdir -p $out
export TMPDIR=/tmp
for i in 1 2; do
( time timeout 2400 python -m pytest tests -q -m gpu -x < /dev/null ) 2>&1 | tail -60 | tee $out/tests_$i.log | tail -6
done
