#!/bin/bash
# tools/gpu_job.sh TAG -- the check of a tree on the GPU box: the GPU tests, the default bench line (+ its detail file), smoke
TAG=${1:-r06_a}
out=gpurun_out/$TAG; mkdir -p $out
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -q -m gpu -x < /dev/null ) 2>&1 | tail -15 | tee $out/tests.log
timeout 1500 python bench.py < /dev/null > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/tests.log; tail -c 2500 $out/bench.json; wc -c $out/bench.json
cp gpurun_out/bench_detail.json $out/bench_detail.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $out/smoke.log
