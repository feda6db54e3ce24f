#!/bin/bash
rm -rf gpurun_out/r03_n gpurun_out/r03_n_*
mkdir -p gpurun_out/r03_final
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r03_final/gpu_tests.txt; cat gpurun_out/r03_final/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 | tee gpurun_out/r03_final/smoke.txt
timeout 1500 bash tools/profile_all.sh r03_n > gpurun_out/r03_final/profile_all.log 2>&1
tail -c 200 gpurun_out/r03_n/bench.json
