#!/bin/bash
# GPU box: build and run the fused attention + o_proj lab (tools/attn_oproj_lab.hip); output -> gpurun_out/
set -e
mkdir -p gpurun_out build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -I qeft_amd/csrc -I tools tools/attn_oproj_lab.hip -o build/attn_oproj_lab > gpurun_out/attn_oproj_lab_build.log 2>&1
timeout -k 10 120 ./build/attn_oproj_lab 32 200 > gpurun_out/attn_oproj_lab.txt 2>&1
cat gpurun_out/attn_oproj_lab.txt
