#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_solar_cycle.py tests/test_driver_outputs.py -m gpu -q 2>&1 | tail -5
