#!/bin/bash
# GPU session: steady-state GEMM numbers (forward v2 tiles vs v3, dX), then the rocprofv3 passes on the 4096^2 forward.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== forward, 128-row tiles (QEFT_GEMM_V3=0)"
QEFT_GEMM_V3=0 timeout -k 10 200 python tools/bench_gemm.py --ms 2048,4096 2>&1 | grep TFLOP || exit 1
echo "== forward, 256x128 tile"
timeout -k 10 200 python tools/bench_gemm.py --ms 1024,2048,4096 2>&1 | grep TFLOP || exit 1
echo "== dX"
timeout -k 10 200 python tools/bench_gemm.py --ms 2048,4096 --bwd 2>&1 | grep TFLOP || exit 1
} | tee gpurun_out/r2_gemm_steady.txt
timeout -k 10 600 bash tools/gpu_prof_gemm.sh > gpurun_out/r2_gemm_prof.txt 2>&1
tail -40 gpurun_out/r2_gemm_prof.txt
