#!/bin/bash
# one GPU call: full GPU suite + smoke + the driver's bench command on the final build
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r02_zm_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r02_zm_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_zm_bench.json 2> gpurun_out/r02_zm_bench.err
python - <<'PY'
import json
for line in open('gpurun_out/r02_zm_bench.json'):
    if line.startswith('{'):
        d=json.loads(line)
        print(d['config']['workload'], d['value'], d['roofline']['frac'], d['parity'])
        for k,w in d['workloads'].items(): print(k, w['value'], w['roofline']['frac'], w['parity']['ok'], w['parity']['max_rel_diff_vs_oracle'])
PY
