#!/bin/bash
# GPU box: build and run the v3 GEMV lab (tools/gemv_v3_lab.hip) and the GEMV parity tests; output -> gpurun_out/
set -e
mkdir -p gpurun_out build
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -I qeft_amd/csrc tools/gemv_v3_lab.hip -o build/gemv_v3_lab > gpurun_out/gemv_v3_lab_build.log 2>&1
timeout -k 10 300 ./build/gemv_v3_lab > gpurun_out/gemv_v3_lab.txt 2>&1
grep -v "^stream" gpurun_out/gemv_v3_lab.txt
timeout -k 10 900 python -m pytest tests/test_gpu_gemv.py tests/test_gpu_gemv_v3.py tests/test_gpu_decode.py tests/test_gpu_engine_7b.py -x -q -m gpu > gpurun_out/gemv_tests.log 2>&1 || true
tail -5 gpurun_out/gemv_tests.log
