#!/bin/bash
# GPU box: build and run the v3 GEMV lab alone (tools/gemv_v3_lab.hip); output -> gpurun_out/gemv_v3_lab.txt
set -e
mkdir -p gpurun_out build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -I qeft_amd/csrc tools/gemv_v3_lab.hip -o build/gemv_v3_lab > gpurun_out/gemv_v3_lab_build.log 2>&1 || { tail -20 gpurun_out/gemv_v3_lab_build.log; exit 1; }
timeout -k 10 400 ./build/gemv_v3_lab > gpurun_out/gemv_v3_lab.txt 2>&1
grep -v "^stream" gpurun_out/gemv_v3_lab.txt
