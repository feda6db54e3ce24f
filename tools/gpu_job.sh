#!/bin/bash
mkdir -p gpurun_out/r03_zf
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
