#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_reference_targets.py -m gpu -q 2>&1 | tail -8
