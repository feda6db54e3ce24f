#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -12
python bench.py --steps 20 --warmup 5 --headline-only 2>&1 | tail -1 | cut -c1-900
