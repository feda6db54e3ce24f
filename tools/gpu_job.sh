#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_spartacus.py tests/test_hip_rrtmg.py -m gpu -x -q -k "spartacus or spectral" 2>&1 | tail -15
