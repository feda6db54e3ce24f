#!/bin/bash
mkdir -p gpurun_out/r03_zy
timeout 900 python -m pytest tests/test_fortran_dropin.py -q -m gpu -k "two_spartacus_regions" -s 2>&1 | grep -E "two regions|passed|failed|FAILED|Error|assert|E  " | tail -30
timeout 900 python -m pytest tests -q -m gpu -k "spartacus" -x 2>&1 | tail -3
for v in inline swlate overlap; do
  E="A=1"; [ $v = swlate ] && E="ECRAD_GEN_SW_LATE=1"; [ $v = overlap ] && E="ECRAD_GEN_OVERLAP=1"
  w=mcica_rrtmg
  env $E timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only --workload $w --ncol 100000 > gpurun_out/r03_zy/${v}_$w.json 2> gpurun_out/r03_zy/${v}_$w.err
  python - $v $w <<'PY'
import json,sys
v,w=sys.argv[1:3]
d=json.loads(open(f"gpurun_out/r03_zy/{v}_{w}.json").read().strip().splitlines()[-1])
s=d["roofline"]["stage_ms"]
print(f"{v:10s} {w:28s} {d['value']:10.0f} col/s {d['ms_per_step']:9.2f} ms  lw {s.get('lw',0):7.2f} sw {s.get('sw',0):7.2f} prep {s.get('prep',0):6.2f}")
PY
done
