#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/r02_zt_bench.json
