#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r02_zu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> gpurun_out/r02_zu_tests.log 2>&1
tail -3 gpurun_out/r02_zu_tests.log
bash tools/profile.sh r02_zu > gpurun_out/r02_zu_profile.log 2>&1
python tools/summarize_prof.py r02_zu > gpurun_out/r02_zu_summary.log 2>&1
tail -3 gpurun_out/r02_zu_summary.log
ls gpurun_out/r02_zu | head
