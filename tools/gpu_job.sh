#!/bin/bash
out=gpurun_out/r04_m; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python tools/small_call_latency.py --host --ncol 80 --threads 1,16,32,64 --contexts 8 2>&1 | grep columns/s | tee -a $out/small.log
timeout 300 python tools/small_call_latency.py --host --ncol 80 --threads 16,32 --contexts 8 --solver McICA 2>&1 | grep columns/s | tee -a $out/small.log
timeout 900 python -m pytest tests/test_hip_pool.py tests/test_hip_tiling.py -q -m gpu -x 2>&1 | tail -3 | tee $out/tests.log
timeout 900 python -m pytest tests/test_fortran_dropin.py -q -m gpu -x -s -k "openmp or zz" 2>&1 | grep -E "OpenMP|passed|failed" | tee $out/dropin_omp.log
timeout 600 python tools/host_mode_rate.py clear_homogeneous_ecckd32 5120 2>&1 | tail -2 | tee $out/host_mode.log
ECRAD_HIP_NO_PIPELINE=1 timeout 600 python tools/host_mode_rate.py clear_homogeneous_ecckd32 5120 2>&1 | tail -1 | tee -a $out/host_mode.log
timeout 600 python tools/host_mode_rate.py tripleclouds_ecckd32 5120 2>&1 | tail -1 | tee -a $out/host_mode.log
ECRAD_HIP_NO_PIPELINE=1 timeout 600 python tools/host_mode_rate.py tripleclouds_ecckd32 5120 2>&1 | tail -1 | tee -a $out/host_mode.log
