#!/bin/bash
mkdir -p gpurun_out/r03_n
python -m pytest tests/test_fortran_dropin.py -m gpu -q -s -k "single_precision or ckdmip" 2>&1 | grep -E "max|passed|failed|single-precision|^E " | cut -c1-400 | tail -30
