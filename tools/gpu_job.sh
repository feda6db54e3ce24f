#!/bin/bash
# one GPU call: the final build -- full GPU suite, smoke, default-run profile (tools/profile.sh: rocprofv3 stats + PMC passes + the bench line)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r02_zs_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r02_zs_tests.log
tools/profile.sh r02_zs --steps 20 --warmup 5 > gpurun_out/r02_zs_profile.log 2>&1
tail -c 400 gpurun_out/r02_zs/bench.json
