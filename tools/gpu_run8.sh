#!/bin/bash
# GPU session: dX v3 (256x128) parity, then steady-state dX timing A/B.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_gemm.py -q -p no:cacheprovider -k "dx or variant or backward" -x > gpurun_out/r2_t8.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error|assert" gpurun_out/r2_t8.log | tail -30
if [ $rc -ne 0 ]; then tail -50 gpurun_out/r2_t8.log; exit $rc; fi
{
echo "== dX, 128-row tiles (QEFT_DX_V3=0)"
QEFT_DX_V3=0 timeout -k 10 200 python tools/bench_gemm.py --ms 2048,4096 --bwd 2>&1 | grep TFLOP || exit 1
echo "== dX, 256x128 tile"
timeout -k 10 200 python tools/bench_gemm.py --ms 2048,4096 --bwd 2>&1 | grep TFLOP || exit 1
} | tee gpurun_out/r2_dx_steady.txt
