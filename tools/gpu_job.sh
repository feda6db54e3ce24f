#!/bin/bash
# one GPU call: shortwave next to longwave for small batches -- full suite twice (races would show as flakiness), latency table
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -3
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
echo "== overlap"; python tools/small_call_latency.py 2>&1 | grep -v amdgpu.ids
echo "== serial";  ECRAD_NO_SPECTRA_OVERLAP=1 python tools/small_call_latency.py 2>&1 | grep -v amdgpu.ids
