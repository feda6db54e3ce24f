#!/bin/bash
mkdir -p gpurun_out/r03_u
O=gpurun_out/r03_u
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_u/bench.json') if l.startswith('{')][0])
print('headline', d['value'], d['ms_per_step'], d.get('parity',{}).get('max_rel_diff_vs_oracle'), d['roofline']['frac'])
for k,v in d.get('workloads',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('parity',{}).get('max_rel_diff_vs_oracle'), v.get('parity',{}).get('ok'))
PY
