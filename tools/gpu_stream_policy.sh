#!/bin/bash
set -e
mkdir -p gpurun_out build
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_policy_lab.hip -o build/stream_policy_lab > gpurun_out/stream_policy_build.log 2>&1
timeout -k 10 300 ./build/stream_policy_lab > gpurun_out/r04_stream_policy.txt 2>&1
cat gpurun_out/r04_stream_policy.txt
