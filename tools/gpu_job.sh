#!/bin/bash
rm -rf gpurun_out/r03_n gpurun_out/r03_n_*
bash tools/profile_all.sh r03_n
