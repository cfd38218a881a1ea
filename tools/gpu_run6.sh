#!/bin/bash
# GPU session: GEMM v3 parity, then ablations (loaders off / compute off) beside the normal run.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_gemm.py -q -p no:cacheprovider -k "gemm or variant" > gpurun_out/r2_t6.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -E "passed|failed|FAILED|Error|assert" gpurun_out/r2_t6.log | tail -30
if [ $rc -ne 0 ]; then exit $rc; fi
for a in 0 1 2; do
  echo "== QEFT_GEMM_ABL=$a" 
  QEFT_GEMM_ABL=$a timeout -k 10 200 python tools/bench_gemm.py --ms 2048 --reps 10 2>&1 | grep TFLOP || exit 1
done | tee gpurun_out/r2_gemm_v3_ablation.txt
timeout -k 10 200 python tools/bench_gemm.py --ms 4096 --reps 10 2>&1 | grep TFLOP
