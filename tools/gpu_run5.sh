#!/bin/bash
# GPU session: GEMM v3 (256x128) parity, then A/B timing against the 128-row tiles.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_gemm.py -q -p no:cacheprovider -k "gemm or variant" > gpurun_out/r2_t5.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/r2_t5.log
grep -E "passed|failed|FAILED|Error|assert" gpurun_out/r2_t5.log | tail -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
QEFT_GEMM_V3=0 timeout -k 10 200 python tools/bench_gemm.py --ms 2048,4096 --reps 10 > gpurun_out/r2_gemm_v2.txt 2>&1
echo "== 128-row tiles"; cat gpurun_out/r2_gemm_v2.txt
timeout -k 10 200 python tools/bench_gemm.py --ms 2048,4096 --reps 10 > gpurun_out/r2_gemm_v3.txt 2>&1
brc=$?
echo "== 256x128 tile"; cat gpurun_out/r2_gemm_v3.txt
exit $brc
