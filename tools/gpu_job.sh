#!/bin/bash
mkdir -p gpurun_out/r03_zx
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r03_zx/gpu_tests.txt
for w in mcica_ecckd32 mcica_rrtmg; do
  for v in shipped static nooverlap; do
    [ $v = nooverlap ] && [ $w = mcica_ecckd32 ] && continue
    E=""; [ $v = static ] && E="ECRAD_GEN_STATIC=1"; [ $v = nooverlap ] && E="ECRAD_NO_GEN_OVERLAP=1"
    env $E timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only --workload $w --ncol 100000 > gpurun_out/r03_zx/${v}_$w.json 2> gpurun_out/r03_zx/${v}_$w.err
    python - $v $w <<'PY'
import json,sys
v,w=sys.argv[1:3]
d=json.loads(open(f"gpurun_out/r03_zx/{v}_{w}.json").read().strip().splitlines()[-1])
s=d["roofline"]["stage_ms"]
print(f"{v:10s} {w:28s} {d['value']:10.0f} col/s {d['ms_per_step']:9.2f} ms  lw {s.get('lw',0):7.2f} sw {s.get('sw',0):7.2f} prep {s.get('prep',0):6.2f}  parity {d.get('parity',{}).get('max_rel_diff_vs_oracle')}")
PY
  done
done
