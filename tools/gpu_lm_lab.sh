#!/bin/bash
set -e
mkdir -p gpurun_out build
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lm_head_lab.hip -o build/lm_head_lab > gpurun_out/lm_head_lab_build.log 2>&1
timeout -k 10 300 ./build/lm_head_lab > gpurun_out/r04_lm_head_lab.txt 2>&1
cat gpurun_out/r04_lm_head_lab.txt
